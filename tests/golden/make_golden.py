"""Generates tests/golden/*.npz with the CPU oracle (oracle/liboracle.so, fdlibm math).

The Rust reference cannot run here (no cargo; entropy-seeded RNG), so these fixtures pin the
ORACLE's output, not the reference's: they guard the oracle against drift (tests/test_golden.py,
CPU) and give the GPU parity tests a committed target.  Regenerate with:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle_ffi as O  # noqa: E402
import small_scenes  # noqa: E402


def main(only_hi=False, only=None):
    for name in ([] if only_hi else (only or small_scenes.NAMES)):
        scene, cam, p = small_scenes.small(name)
        osc = O.OracleScene(scene)
        img, cnt = osc.render(cam, p, threads=0, counters=True)
        # a few closest-hit records along camera rays of sample 0
        rays = [O.camera_ray(cam, p, x, y, 0) for y in range(0, p.height, 4) for x in range(0, p.width, 4)]
        o = np.array([r[0] for r in rays])
        d = np.array([r[1] for r in rays])
        t, n, obj = osc.closest_hit(o, d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, ray_o=o, ray_d=d, hit_t=t, hit_n=n,
                            hit_obj=obj, counters=np.array([cnt[k] for k in sorted(cnt)], dtype=np.int64),
                            counter_names=np.array(sorted(cnt)))
        print(name, img.shape, "mean", img.mean(axis=0), "hits", int((obj >= 0).sum()), "/", len(obj))
    for name in ([] if only else small_scenes.HI_NAMES):  # image only (~10^6 samples each)
        scene, cam, p = small_scenes.small(name)
        img = O.OracleScene(scene).render(cam, p, threads=0)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img)
        print(name, img.shape, "mean", img.mean(axis=0))


if __name__ == "__main__":
    main(only_hi="--hi" in sys.argv, only=[a for a in sys.argv[1:] if not a.startswith("--")] or None)
