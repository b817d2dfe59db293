import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import small_scenes
from rpt_amd import GpuScene, _abi, make_params
scene, cam, p = small_scenes.small("fractal_teapots")
for env in ({}, {"RPTGPU_DEEP_DEPTH": "1"}, {"RPTGPU_DEEP_DEPTH": "1", "RPTGPU_NEST_TRACE": "0"}, {"RPTGPU_NEST_TRACE": "0"}, {"RPTGPU_NEST_PER_TREE": "0"}):
    for k in ("RPTGPU_DEEP_DEPTH", "RPTGPU_NEST_PER_TREE", "RPTGPU_NEST_TRACE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.time(); g = GpuScene(scene, 0); t1 = time.time()
    for flags in (_abi.RPT_FLAG_WAVEFRONT, 0):
        t2 = time.time()
        g.render_batch(cam, make_params(p.width, p.height, p.max_bounces, p.iterations, p.exposure_value, p.seed, flags=flags))
        t3 = time.time()
        g.render_batch(cam, make_params(p.width, p.height, p.max_bounces, p.iterations, p.exposure_value, p.seed, flags=flags))
        t4 = time.time()
        print(env, "flags", flags, "create %.3f s  render#1 %.3f s  render#2 %.3f s" % (t1 - t0, t3 - t2, t4 - t3), flush=True)
    g.close()
