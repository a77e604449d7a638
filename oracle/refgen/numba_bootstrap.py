"""Makes conda's numba 0.54.1 importable under the pip-upgraded numpy 1.26.4
(SURVEY.md Appendix A).  Used ONLY by oracle/refgen/*.py in the build container
with /opt/conda/bin/python3.9 to run the unmodified reference from
/root/reference and emit golden vectors.  Never imported by product code and
never needed on the GPU box."""
import sys, types, warnings
warnings.filterwarnings('ignore')
fake = types.ModuleType('numba.np.ufunc._internal')   # C ext that fails to init; only @vectorize needs it
class _DUFunc(object):
    def __init__(self, *a, **k):
        raise NotImplementedError
fake._DUFunc = _DUFunc
fake.PyUFunc_None, fake.PyUFunc_Zero, fake.PyUFunc_One, fake.PyUFunc_ReorderableNone = -1, 0, 1, -2
def _fromobject(*a, **k):
    raise NotImplementedError
fake.fromobject = _fromobject
sys.modules['numba.np.ufunc._internal'] = fake
import numpy as _np
from numpy.core._machar import MachAr as _MachAr
_np.MachAr = _MachAr                                    # removed in numpy 1.24
for _n, _t in [('bool', bool), ('int', int), ('float', float), ('complex', complex),
               ('object', object), ('str', str)]:
    if _n not in _np.__dict__:
        setattr(_np, _n, _t)
_real = _np.__version__
_np.__version__ = '1.20.3'                              # numba's version gate
try:
    import numba  # noqa: F401
finally:
    _np.__version__ = _real
