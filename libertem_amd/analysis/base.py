"""
Analysis glue: parameters -> UDF -> named results.  Subset of the reference's
libertem.analysis.base (analysis/base.py:17-211); visualisation (`visualized`) is out of scope,
`raw_data` and the result names/keys are kept.
"""
import numpy as np


class AnalysisResult:
    def __init__(self, raw_data, visualized=None, title="", desc="", key="",
                 include_in_download=True):
        self.include_in_download = include_in_download
        self.raw_data = raw_data
        self._visualized = visualized
        self.title = title
        self.desc = desc
        self.key = key

    @property
    def visualized(self):
        raise NotImplementedError("visualisation is outside the scope of libertem_amd")

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.raw_data)
        return a if dtype is None else a.astype(dtype)

    def __repr__(self):
        shape = getattr(self.raw_data, 'shape', None)
        return f"<AnalysisResult: {self.key} shape={shape}>"


class AnalysisResultSet:
    def __init__(self, results, raw_results=None):
        self._results = results
        self.raw_results = raw_results

    @property
    def results(self):
        if callable(self._results):
            self._results = self._results()
        return self._results

    def __repr__(self):
        return repr(self.results)

    def __getattr__(self, k):
        if k.startswith('_'):
            raise AttributeError(k)
        for result in self.results:
            if result.key == k:
                return result
        raise AttributeError("result with key '%s' not found, have: %s" % (
            k, ", ".join([r.key for r in self.results])))

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return self.results[k]

    def __len__(self):
        return len(self.results)

    def keys(self):
        return [r.key for r in self.results]

    def __iter__(self):
        return iter(self.results)


class Analysis:
    TYPE = 'UDF'
    registry = {}

    def __init_subclass__(cls, id_=None, **kwargs):
        super().__init_subclass__(**kwargs)
        if id_ is not None:
            cls.registry[id_] = cls

    def __init__(self, dataset, parameters):
        self.dataset = dataset
        self.parameters = self.get_parameters(parameters)
        self.parameters.update(parameters)

    def get_parameters(self, parameters):
        return dict(parameters)

    def get_udf(self):
        raise NotImplementedError()

    def get_roi(self):
        return None

    def get_udf_results(self, udf_results, roi, damage):
        raise NotImplementedError()

    def need_rerun(self, old_params, new_params):
        return True

    def get_complex_results(self, job_result, key_prefix, title, desc, damage, default_lin=True):
        """Result list for complex data; the magnitude keeps key=key_prefix for compatibility
        (analysis/base.py:146-203)."""
        magn = np.abs(job_result)
        return [
            AnalysisResult(raw_data=magn, key=key_prefix if default_lin else f'{key_prefix}_lin',
                           title="%s [magn]" % title, desc="%s [magn]" % desc),
            AnalysisResult(raw_data=magn, key=f'{key_prefix}_log' if default_lin else key_prefix,
                           title="%s [log(magn)]" % title, desc="%s [log(magn)]" % desc),
            AnalysisResult(raw_data=job_result.real, key="%s_real" % key_prefix,
                           title="%s [real]" % title, desc="%s [real]" % desc),
            AnalysisResult(raw_data=job_result.imag, key="%s_imag" % key_prefix,
                           title="%s [imag]" % title, desc="%s [imag]" % desc),
            AnalysisResult(raw_data=np.angle(job_result), key="%s_angle" % key_prefix,
                           title="%s [angle]" % title, desc="%s [angle]" % desc),
            AnalysisResult(raw_data=job_result, key="%s_complex" % key_prefix,
                           title="%s [complex]" % title, desc="%s [complex]" % desc),
        ]


BaseAnalysis = Analysis
