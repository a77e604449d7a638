"""CPU: the C-ABI library loads and exports every symbol include/shepseg_hip.h declares, and the
product path fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'shepseg_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(shp_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_header_symbols():
    from pyshepseg_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert 'shp_segment_tile' in names and 'shp_ctx_create' in names
    for n in names:
        assert hasattr(L, n), 'libshepseg_hip.so does not export %s' % n
    assert set(_lib._SIGS) == set(names)
    assert L.shp_version() >= 100


def test_no_cpu_fallback_without_gpu():
    from pyshepseg_amd import _lib, shepseg
    if _lib.lib().shp_device_count() > 0:
        pytest.skip('a GPU is present')
    img = np.zeros((3, 8, 8), dtype=np.uint16)
    with pytest.raises(_lib.ShepsegHipError):
        shepseg.doShepherdSegmentation(img, kmeansObj=shepseg.KMeansModel(np.zeros((2, 3))))


def test_product_never_imports_oracle():
    pk = os.path.join(ROOT, 'pyshepseg_amd')
    for dirpath, _d, files in os.walk(pk):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
                assert 'liboracle' not in src, f


def test_walker_assembly_header_is_the_generators_output(tmp_path):
    """pyshepseg_amd/csrc/dfs_walk4_asm.h is generated: the committed file must be what tools/gen_dfs_walk4.py writes
    today (a hand edit of either would otherwise go unnoticed until the next regeneration)."""
    import subprocess
    import sys
    out = tmp_path / 'dfs_walk4_asm.h'
    env = dict(os.environ, DFS_WALK_OUT=str(out))
    env.pop('DFS_WALK_RUNS', None)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'gen_dfs_walk4.py')], env=env,
                          stdout=subprocess.DEVNULL)
    committed = open(os.path.join(ROOT, 'pyshepseg_amd', 'csrc', 'dfs_walk4_asm.h')).read()
    assert out.read_text() == committed
