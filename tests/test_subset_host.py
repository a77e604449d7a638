"""CPU: argument checking of pyshepseg_amd.subset.subsetImage (reference subset.py:86-88,
:121-123, :168-170) -- raised before anything touches the GPU."""
import numpy as np
import pytest


def test_subset_errors():
    from pyshepseg_amd import subset
    seg = np.zeros((50, 60), dtype=np.uint32)
    with pytest.raises(subset.PyShepSegSubsetError, match='not within input image'):
        subset.subsetImage(seg, None, 10, 10, 60, 10)
    with pytest.raises(subset.PyShepSegSubsetError, match='mask should match'):
        subset.subsetImage(seg, None, 0, 0, 20, 10, maskImage=np.ones((10, 21), np.uint8))
    with pytest.raises(subset.PyShepSegSubsetError, match='No valid data'):
        subset.subsetImage(seg, None, 0, 0, 20, 10)
    seg[:] = 7
    with pytest.raises(subset.PyShepSegSubsetError, match='No valid data'):
        subset.subsetImage(seg, None, 0, 0, 20, 10, maskImage=np.zeros((10, 20), np.uint8))
