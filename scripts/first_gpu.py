import sys, time
sys.path.insert(0, '.')
import numpy as np
from rpt_amd import scenes, make_params, GpuScene, device_count
from oracle import oracle_ffi as O
print('devices', device_count())
import subprocess
print(subprocess.run('nproc; lscpu | grep "Model name"; rocminfo | grep -m2 gfx', shell=True, capture_output=True, text=True).stdout)
for name, res in [('sphere',(160,90)),('cornell',(160,90)),('fractal_spheres',(160,90))]:
    sc, cam, d = scenes.SCENES[name]()
    g = GpuScene(sc, 0)
    osc = O.OracleScene(sc)
    p = make_params(res[0], res[1], d['max_bounces'], 4, seed=3)
    # closest-hit parity on camera rays
    rays = [O.camera_ray(cam, p, x, y, 0) for y in range(0, res[1], 3) for x in range(0, res[0], 3)]
    o = np.array([r[0] for r in rays]); dd = np.array([r[1] for r in rays])
    t0, n0, ob0 = osc.closest_hit(o, dd)
    t1, n1, ob1 = g.closest_hit(o, dd)
    print(name, 'closest_hit: t bit-equal', (t0 == t1).mean(), 'obj equal', (ob0 == ob1).mean(), 'n equal', (n0 == n1).all(axis=1).mean())
    t = time.time(); img = g.render_batch(cam, p); tg = time.time() - t
    t = time.time(); ref = osc.render(cam, p, threads=0); tc = time.time() - t
    tol = 1e-9 * np.maximum(1, np.abs(ref))
    ok = (np.abs(img - ref) <= tol).all(axis=1)
    print(name, 'render: gpu %.3fs cpu %.3fs' % (tg, tc), 'frac ok', ok.mean(), 'bit-equal', (img == ref).all(axis=1).mean(), 'max diff', np.abs(img-ref).max(), 'nan', np.isnan(img).sum())
    bad = np.where(~ok)[0][:5]
    for b in bad: print('  bad pixel', b, img[b], ref[b])
# quick throughput probe at 1080p cornell
sc, cam, d = scenes.cornell()
g = GpuScene(sc, 0)
p = make_params(1920, 1080, 8, 4, seed=1, flags=1)
g.render_batch(cam, p)
g.reset_stats()
t = time.time(); g.render_batch(cam, p); dt = time.time() - t
s = g.stats()
print('cornell 1080p 4spp B=8: %.3fs -> %.2f Msamples/s' % (dt, 1920*1080*4/dt/1e6))
print('kernel ms', list(s.kernel_ms)[:5], 'launches', list(s.kernel_launches)[:5], 'extend rays', s.extend_rays, 'shadow rays', s.shadow_rays)
