"""The committed fixtures under tests/golden/ (made by tests/golden/make_golden.py with the CPU
oracle) are reproduced bit for bit by the oracle here — the oracle does not drift — and by the
HIP path on the GPU (tests/test_gpu_parity.py)."""
import os

import numpy as np
import pytest

import small_scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.mark.parametrize("name", small_scenes.NAMES)
def test_oracle_reproduces_golden(oracle, name):
    z = load(name)
    scene, cam, p = small_scenes.small(name)
    osc = oracle.OracleScene(scene)
    img, cnt = osc.render(cam, p, threads=2, counters=True)
    assert (img == z["image"]).all()
    t, n, obj = osc.closest_hit(z["ray_o"], z["ray_d"])
    assert (t == z["hit_t"]).all() and (n == z["hit_n"]).all() and (obj == z["hit_obj"]).all()
    assert [cnt[k] for k in sorted(cnt)] == z["counters"].tolist()
    assert np.isfinite(img).all() and (img >= 0).all()


@pytest.mark.parametrize("name", small_scenes.HI_NAMES)
def test_oracle_reproduces_high_sample_golden(oracle, name):
    # 256x144 at 32 spp (~10^6 samples per scene): image only
    scene, cam, p = small_scenes.small(name)
    img = oracle.OracleScene(scene).render(cam, p, threads=0)
    ref = load(name)["image"]
    assert (img == ref).all() and np.isfinite(img).all()


def test_golden_covers_every_device_feature():
    z = {n: load(n) for n in small_scenes.NAMES}
    names = z["coverage"]["counter_names"].tolist()
    cov = dict(zip(names, z["coverage"]["counters"].tolist()))
    for k in ("n_tri", "n_sphere", "n_plane", "n_cube", "n_inst", "n_inner", "n_leaf", "shadow_rays", "misses"):
        assert cov[k] > 0, k
    frac = dict(zip(names, z["fractal_spheres"]["counters"].tolist()))
    assert frac["n_inner"] > 0 and frac["n_sphere"] > 0 and frac["n_tri"] == 0
    dr = dict(zip(names, z["dragon"]["counters"].tolist()))
    assert dr["n_tri"] > 0 and dr["n_inner"] > dr["n_leaf"] > 0
