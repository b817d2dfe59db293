"""A shadow ray that ends in a NaN hit.  MonomialSurface::intersect (monomial_surface.rs:50-104) reports a hit at
time NaN for an exactly vertical ray over a corner of its bounding box (the Newton step divides by -0); for a SHADOW
ray `closest_hit.unwrap() > dist_to_light` (renderer.rs:197) is then false and the light is not added.  The device
answers "record.time > stop", which is false for NaN as well (kernels/traversal.inc visible()); it used to answer
"!(time <= stop)", which is true.  The scene makes a whole patch of the image depend on that decision."""
import numpy as np
import pytest

import rpt_amd
from rpt_amd import GpuScene, _abi, make_params

pytestmark = pytest.mark.gpu


def nan_shadow_scene(with_mesh=False):
    from rpt_amd import scenes
    s = rpt_amd.Scene()
    # a ceiling patch over the (+x, +z) corner of the surface's box, facing DOWN: 5 < 2 (x^2 + z^2)^2 on all of it, so a
    # vertical ray from it starts "below" the surface's extension and takes the maximise branch (:39-64)
    s.add(rpt_amd.Object(rpt_amd.polygon([(0.9, 5.0, 0.9), (1.0, 5.0, 0.9), (1.0, 5.0, 1.0), (0.9, 5.0, 1.0)]))
          .material(rpt_amd.Material.diffuse((0.8, 0.7, 0.6))))
    s.add(rpt_amd.Object(rpt_amd.monomial_surface(2.0, 4.0)).material(rpt_amd.Material.diffuse((0.5, 0.5, 0.9))))
    if with_mesh:  # a deep tree elsewhere, so that the per-tree (object by object) pipeline runs the visibility queries
        s.add(rpt_amd.Object(rpt_amd.Mesh(scenes.knot_mesh(64, 12)).scale((0.2, 0.2, 0.2)).translate((-3.0, 1.0, -3.0)))
              .material(rpt_amd.Material.diffuse((0.6, 0.6, 0.6))))
    s.add(rpt_amd.Light.Directional((2.0, 2.0, 2.0), (0.0, 1.0, 0.0)))  # wi = -normalize(dir) = (0, -1, 0) exactly
    s.add(rpt_amd.Light.Ambient((0.05, 0.05, 0.05)))
    cam = rpt_amd.Camera.look_at((0.95, 3.2, 0.95), (0.95, 5.0, 0.9501), (0.0, 0.0, 1.0), 0.12)
    return s, cam


@pytest.mark.parametrize("with_mesh", [False, True])
def test_a_nan_shadow_hit_occludes_like_the_reference(with_mesh, oracle, monkeypatch):
    if with_mesh:
        monkeypatch.setenv("RPTGPU_DEEP_DEPTH", "1")
    scene, cam = nan_shadow_scene(with_mesh)
    p = make_params(48, 48, 1, 4, seed=11)
    ref = oracle.OracleScene(scene).render(cam, p, threads=0)
    # the oracle itself: on the patch, the directional light is never added (only ambient x albedo) ...
    o = oracle.OracleScene(scene)
    hit, t, n = oracle.shape_intersect(rpt_amd.monomial_surface(2.0, 4.0), (0.95, 5.0, 0.95), (0.0, -1.0, 0.0))
    assert hit and np.isnan(t)
    centre = ref.reshape(48, 48, 3)[24, 24]
    assert np.allclose(centre, 0.05 * np.array([0.8, 0.7, 0.6]), rtol=0.3), centre  # ambient (+ one dim bounce), not 2 * bsdf
    # ... and every device pipeline agrees bit for bit
    for flags in (0, _abi.RPT_FLAG_PERSISTENT, _abi.RPT_FLAG_WAVEFRONT):
        g = GpuScene(scene, 0)
        img = g.render_batch(cam, make_params(48, 48, 1, 4, seed=11, flags=flags))
        g.close()
        same = (img == ref) | (np.isnan(img) & np.isnan(ref))
        assert same.all(), (flags, np.flatnonzero(~same.all(axis=1))[:8], img[~same.all(axis=1)][:3], ref[~same.all(axis=1)][:3])


def test_gpu_against_the_oracle_built_with_the_system_libm(oracle):
    """Oracle and kernels share include/rpt_math.h, so "GPU == oracle, bit for bit" says nothing about those six
    functions.  This is the independent leg: the same C2 frame from the device and from the oracle built against
    glibc's libm (liboracle_sysm.so).  A 1-ulp difference in exp / ln / atan / sincos flips a branch on a small
    share of samples, so the comparison is the north star's tolerance form: per-channel |delta| <= 1e-9 max(1, |x|) on
    the bulk of the pixels, and the same mean."""
    import ctypes as C
    from rpt_amd import scenes
    scene, cam, _ = scenes.cornell()
    W, H, B, spp = 256, 144, 8, 32
    p = make_params(W, H, B, spp, seed=5)
    g = GpuScene(scene, 0)
    img = g.render_batch(cam, p)
    g.close()
    L = oracle.sysm_lib()
    desc, keep = scene.lower()
    camc = cam.lower()
    h = C.c_void_p()
    assert L.oracle_scene_create(C.byref(desc), C.byref(h)) == 0
    ref = np.empty((W * H, 3))
    assert L.oracle_render(h, C.byref(camc), C.byref(p), 0, ref.ctypes.data_as(C.POINTER(C.c_double)), None) == 0
    L.oracle_scene_destroy(h)
    close = (np.abs(img - ref) <= 1e-9 * np.maximum(1.0, np.abs(ref))).all(axis=1)
    frac = close.mean()
    print("GPU vs glibc-libm oracle, C2 %dx%d %d spp: %.2f %% of pixels within 1e-9, mean %.6f vs %.6f" % (W, H, spp, 100 * frac, img.mean(), ref.mean()))
    assert frac >= 0.95, frac
    assert abs(img.mean() - ref.mean()) / ref.mean() < 2e-3
    # and against the fdlibm oracle (the one that shares rpt_math.h) the same frame is exact
    exact = oracle.OracleScene(scene).render(cam, p, threads=0)
    assert (img == exact).all()
