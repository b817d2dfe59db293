#!/usr/bin/env python
"""One-off run of the asset-driven example programs on the reference's OWN asset files (wine_glass.obj, pegasus.zip,
teapot.obj, rustacean.obj, cylinder.stl), which are not redistributed in this repository: point $RPT_ASSETS at a directory
that holds them (the reference's examples/).  For every scene: import through rpt_amd.io, flatten + kd build + upload
(scene_create), throughput of rptgpu_render_batch at the example's / BASELINE's frame size (D2H included), and a bit-for-bit
comparison with the oracle on 1/`parts` of the tiles.  Committed output: profiles/r02_real_assets.txt.

    RPT_ASSETS=/path/to/reference/examples python scripts/real_assets.py [spp] [parts]
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
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L, how = O.baseline_lib(native=True)
print("assets from %s; oracle build: %s; parity on 1/%d of the tiles at 2 spp; throughput at %d spp"
      % (os.environ.get("RPT_ASSETS"), how, parts, spp))


def need(name):
    t0 = time.time()
    m = scenes.load_asset(name)
    if m is None:
        print("%s: not found in $RPT_ASSETS, skipped" % name)
        return None
    print("%-16s %7d triangles, imported in %.1f s" % (name, len(m.triangles), time.time() - t0))
    return m


def bounds(mesh):
    t = np.asarray(mesh.triangles)[:, :9].reshape(-1, 3)
    return t.min(axis=0), t.max(axis=0)


def run(label, scene, cam, cfg, spp_run):
    W, H, B = cfg["width"], cfg["height"], cfg["max_bounces"]
    ev = cfg.get("exposure_value", 0.0)
    t0 = time.time()
    g = GpuScene(scene, 0)
    t_create = time.time() - t0
    p = make_params(W, H, B, 2, seed=0xA55E7 + len(label), tile=(32, 8), part=(5 % parts, parts), exposure_value=ev)
    ref = O.OracleScene(scene, L).render(cam, p, threads=0)
    img = g.render_batch(cam, p)
    same = (img.view(np.int64) == ref.view(np.int64)) | (np.isnan(img) & np.isnan(ref))
    n = int((ref != 0).any(axis=1).sum()) * 2
    full = make_params(W, H, B, spp_run, seed=0xBEEF, exposure_value=ev)
    g.render_batch(cam, make_params(W, H, B, 1, seed=1, exposure_value=ev))   # warm-up
    best = 1e30
    for _ in range(2):
        t0 = time.time()
        g.render_batch(cam, full)
        best = min(best, time.time() - t0)
    g.close()
    print("%-34s %4dx%-4d B=%-2d  scene_create %.2f s  %8.1f Msamples/s (%d spp, %.2f s)  parity: %d samples, %d differing  %s"
          % (label, W, H, B, t_create, W * H * spp_run / best / 1e6, spp_run, best, n, int((~same).sum()),
             "BIT-EQUAL" if same.all() else "MISMATCH"))
    assert same.all(), label


glass = need("wine_glass.obj")
if glass is not None:
    run("wine_glass.rs (C5, the real mesh)", *scenes.wine_glass(mesh=glass), spp)
horse = need("pegasus.obj")
if horse is not None:
    lo, hi = bounds(horse)
    placed = horse.scale((2.0, 2.0, 2.0)).translate((0.0, -1.0 - 2.0 * lo[1], 0.0))   # SURVEY §8d, stand-in (ii)
    run("dragon.rs with pegasus.obj (C3)", *scenes.dragon(shape=placed), spp)
    run("pegasus.rs", *scenes.pegasus(mesh=horse), 4 * spp)
crab = need("rustacean.obj")
if crab is not None:
    run("rustacean.rs", *scenes.rustacean(mesh=crab), 4 * spp)
pot = need("teapot.obj")
if pot is not None:
    run("teapot.rs", *scenes.teapot(mesh=pot), 16 * spp)
    run("metal.rs", *scenes.metal(mesh=pot), 4 * spp)
    s, c, d = scenes.fractal_teapots(mesh=pot)
    run("fractal_teapots.rs", s, c, d, 16 * spp)
    d8 = dict(d, max_bounces=8)
    run("fractal_teapots.rs at 8 bounces", s, c, d8, 4 * spp)
tube = need("cylinder.stl")
if tube is not None:
    run("cylinder.rs", *scenes.cylinder(mesh=tube), 16 * spp)
