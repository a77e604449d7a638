"""Import-only GDAL stand-in so the reference's tiling/tilingstats modules can be
imported for their pure-numpy/numba static helpers.  Any real I/O call raises."""
import types, sys
class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name == 'UseExceptions':
            return lambda *a, **k: None
        if name.startswith(('GDT_', 'GFT_', 'GFU_', 'GA_')):
            return hash(name) % 1000
        if name == 'Dataset':
            return type('Dataset', (), {})
        def _fail(*a, **k):
            raise RuntimeError('GDAL stub: %s' % name)
        return _fail
gdal = _Stub('osgeo.gdal'); osr = _Stub('osgeo.osr'); gdal_array = _Stub('osgeo.gdal_array')
sys.modules.update({'osgeo.gdal': gdal, 'osgeo.osr': osr, 'osgeo.gdal_array': gdal_array})
