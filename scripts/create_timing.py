"""Where rptgpu_scene_create's time goes: RPTGPU_PRINT_CREATE=1 python scripts/create_timing.py [scene ...]
Each scene is created three times in one process (the first handle of a process pays one-off costs), then rendered
once at a tiny size and created again (after the first launch the code objects are loaded)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rpt_amd  # noqa: E402
from rpt_amd import make_params, scenes  # noqa: E402

for name in sys.argv[1:] or ["cornell", "dragon"]:
    scene, cam, cfg = scenes.SCENES[name]()
    t0 = time.perf_counter()
    desc, keep = scene.lower()
    print("%s: python scene.lower() %.1f ms" % (name, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
    for i in range(3):
        t0 = time.perf_counter()
        g = rpt_amd.GpuScene(scene, 0)
        print("%s: create #%d %.1f ms (incl. lower)" % (name, i, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
        if i == 1:
            t0 = time.perf_counter()
            g.render_batch_reduce(cam, make_params(64, 36, 2, 1), root=0, out=np.empty(64 * 36 * 3, dtype=np.float32))
            print("%s: first tiny render %.1f ms" % (name, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
            t0 = time.perf_counter()
            g.render_batch_reduce(cam, make_params(64, 36, 2, 1), root=0, out=np.empty(64 * 36 * 3, dtype=np.float32))
            print("%s: second tiny render %.1f ms" % (name, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
        t0 = time.perf_counter()
        g.close()
        print("%s: destroy %.1f ms" % (name, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
