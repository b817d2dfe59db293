#!/usr/bin/env python
"""One-off large parity sweep on the GPU box (the committed output is profiles/r02_parity_sweep.txt): every BASELINE
config at its FULL frame size, default pipeline and environment; the oracle (uninstrumented build, all granted host
cores) renders 1/8 of the tiles at 16 spp, the GPU renders the same part; every pixel is compared bit for bit.
The -m gpu tests do the same on 1/64..1/256 of the tiles at 2 spp; this is the same check on 60-250x more samples.

    python scripts/parity_sweep.py [spp] [parts] [part] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi as O  # noqa: E402
from rpt_amd import GpuScene, make_params, scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
part = int(sys.argv[3]) if len(sys.argv) > 3 else 3 % parts
seed0 = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0xABCDE
L, how = O.baseline_lib(native=True)
print("oracle build: %s; %d spp on part %d of %d of the tiles, seeds from %#x" % (how, spp, part, parts, seed0))
total = 0
for name in ("sphere", "cornell", "dragon", "fractal_spheres", "glass", "wine_glass", "room23", "fractal_teapots"):
    scene, cam, cfg = scenes.SCENES[name]()
    if name == "fractal_teapots":  # the example renders with 0 bounces; with 8, shadow and bounce rays walk the nests too
        cfg = dict(cfg, max_bounces=8)
    W, H, B = cfg["width"], cfg["height"], cfg["max_bounces"]
    p = make_params(W, H, B, spp, seed=seed0 + len(name), tile=(32, 8), part=(part, parts))
    t0 = time.time()
    ref = O.OracleScene(scene, L).render(cam, p, threads=0)
    t1 = time.time()
    g = GpuScene(scene, 0)
    img = g.render_batch(cam, p)
    t2 = time.time()
    g.close()
    same = (img.view(np.int64) == ref.view(np.int64)) | (np.isnan(img) & np.isnan(ref))
    n = int((ref != 0).any(axis=1).sum()) * spp
    total += n
    print("%-16s %dx%d B=%-2d: %9d samples compared, %d differing values, oracle %.1f s, GPU %.2f s  %s"
          % (name, W, H, B, n, int((~same).sum()), t1 - t0, t2 - t1, "BIT-EQUAL" if same.all() else "MISMATCH"))
    assert same.all(), name
print("total: %d samples, all bit-equal" % total)
