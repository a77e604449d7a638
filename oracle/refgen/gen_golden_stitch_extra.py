"""Search small synthetic rasters for stitch runs in which the reference leaves EMPTY ids behind
and recodes labelled pixels to 0 (the quirks SURVEY 8e.2 asks to reproduce, not fix), and save
the first hit as a golden (build container only):
    /opt/conda/bin/python3.9 oracle/refgen/gen_golden_stitch_extra.py
Everything that computes is the reference's (through gen_golden.stitch_case's harness)."""
import os
import sys
import numpy as np

import refenv  # noqa: F401
import gen_golden
from oracle import oracle

import io, contextlib
found = 0
have = {'zeros': 0, 'empties': 0}
for seed in range(40, 1200):
    rng = np.random.RandomState(seed)
    nr, nc = int(rng.randint(150, 260)), int(rng.randint(150, 260))
    img = oracle.synthimg(seed, 3, nr, nc)
    if seed % 2:
        img[:, rng.rand(nr, nc) < 0.04] = 65535
    tile, ov = int(rng.choice([48, 64, 80])), int(rng.choice([16, 24, 32]))
    saved = {}
    gen_golden.save = lambda name, **arrs: saved.update(arrs)          # capture instead of writing
    with contextlib.redirect_stdout(io.StringIO()):
        gen_golden.stitch_case('x', img, tile, ov, 6, int(rng.randint(20, 60)), 65535 if seed % 2 else None,
                               bool(rng.randint(0, 2)))
    hist, mosaic = saved['hist'], saved['mosaic']
    nonnull = ~(img == 65535).any(axis=0) if seed % 2 else np.ones(mosaic.shape, bool)
    empties = int((hist[1:] == 0).sum())
    zeros = int(((mosaic == 0) & nonnull).sum())
    kind = None
    if empties >= 1 and have['empties'] < 1:
        kind = 'empties'
    elif zeros >= 500 and empties == 0 and have['zeros'] < 1:
        kind = 'zeros'
    if kind:
        have[kind] += 1
        saved['stack'] = np.array(refenv.STACK)
        np.savez_compressed(os.path.join(gen_golden.OUT, 'stitch_quirk_%s.npz' % kind), **saved)
        print('saved stitch_quirk_%s: seed %d %dx%d tile %d ov %d empties %d labelled->0 pixels %d'
              % (kind, seed, nr, nc, tile, ov, empties, zeros), flush=True)
        found += 1
        if have['empties'] and have['zeros']:
            break
print('found', found)
