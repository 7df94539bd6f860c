"""
libertem_amd -- MI355X-native mask application / virtual detectors behind LiberTEM's UDF API.

Only the hot path `ApplyMasksUDF` / CoM / sum / radial-Fourier is implemented.
"""
__version__ = "0.1.0"
