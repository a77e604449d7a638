"""A minimal in-memory stand-in for the parts of ``osgeo`` (GDAL's Python bindings) that
pyshepseg_amd touches, so that its GDAL-facing branches run in an image without GDAL.  Test
infrastructure only (tests put this directory on sys.path); nothing in pyshepseg_amd knows about it.
"Files" live in gdal.REGISTRY (path -> Dataset); every call a test may want to inspect is appended to
gdal.CALLS as a tuple."""
