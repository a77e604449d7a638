"""pyshepseg_amd -- MI355X-native drop-in for the pyshepseg segmentation hot path.

``shepseg``      per-tile Shepherd segmentation (k-means -> clump -> elimination) on HIP
``tiling``       tiled driver + cross-tile stitch
``tilingstats``  per-segment statistics
"""
__version__ = '0.1.0'
