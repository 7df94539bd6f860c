"""
oracle/ -- CPU restatement of the reference's ApplyMasksUDF / virtual-detector hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import anything from here, and only as the *checker*
(or the timed CPU baseline), never as the thing that is shipped or measured as the GPU path.
The product (`libertem_amd`) never imports `oracle` and fails loudly without its HIP library.

Every function cites the reference lines (relative to /root/reference/) it restates.

Pinning: the restatement is checked against golden vectors produced by running the REAL
Python reference in the build container (tests/golden/generate_golden.py, outputs committed as
tests/golden/*.npz) in `tests/test_oracle_golden.py`.  Parity is therefore PINNED for: dense
ApplyMasksUDF (dtype matrix, tile shapes, partitioning), SumUDF, SumSigUDF, CoMUDF, COMAnalysis
post-processing, RadialFourierAnalysis (dense complex64), all mask factories, `rmatmul`
(CSR/CSC), the tiling negotiation and detector corrections (CorrectionSet on Sum/SumSig/
ApplyMasks UDFs, RepairDescriptor tables, correct_dot_masks, tile-shape adjustment) and the
byte-order decoders (io/dataset/base/decode.py, unsigned dtype pairs of the reference's own test).  Not pinned through the import (third-party pydata
`sparse` is absent): construction of *sparse* mask stacks; there the oracle follows
common/container.py:33-71 + masks.py:290-353 and is anchored on `rmatmul` + dense radial_bins.
"""
from . import masks, tiling, path, corrections, decode, mib  # noqa: F401
