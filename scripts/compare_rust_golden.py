#!/usr/bin/env python
"""Pins the CPU oracle to the REFERENCE: compares frames dumped by the patched reference
(`cargo run --release --features philox --example dump_golden -- DIR`, see rust/rpt.patch and
rust/rpt_additions/dump_golden.rs) with the oracle's render of the same configuration and seed.

    python scripts/compare_rust_golden.py DIR [--save-fixtures]

Expected: bit-equal on almost every pixel; the oracle evaluates exp/ln/atan/sin_cos/acos/atan2 with the fdlibm
restatement of include/rpt_math.h while Rust calls the platform libm (<= 1 ulp apart on ~10 % of arguments), so a
small fraction of pixels may differ in the last bits or by one re-rolled branch.  With --save-fixtures the
reference's frames are stored under tests/golden/ref_<name>.npz, after which tests/test_golden.py holds the oracle
to them and the "parity unpinned" note in DESIGN.md can go.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi as O  # noqa: E402
from rpt_amd import make_params, scenes  # noqa: E402

SCENES = {"sphere": scenes.sphere_scene, "cornell": scenes.cornell}


def main():
    d = sys.argv[1]
    save = "--save-fixtures" in sys.argv
    worst = 0.0
    for name, factory in SCENES.items():
        txt = os.path.join(d, name + ".txt")
        if not os.path.exists(txt):
            print(name, ": no dump")
            continue
        kv = dict(line.split() for line in open(txt))
        w, h, b, n, seed = (int(kv[k]) for k in ("width", "height", "max_bounces", "iterations", "seed"))
        ref = np.fromfile(os.path.join(d, name + ".f64"), dtype="<f8").reshape(h * w, 3)
        scene, cam, _ = factory()
        p = make_params(w, h, b, n, seed=seed)
        for lib, label in ((O.lib(), "oracle (fdlibm restatement)"),
                           (O._load(os.path.join(ROOT, "oracle", "liboracle_sysm.so")), "oracle (system libm)")):
            img = O.OracleScene(scene, lib).render(cam, p, threads=0)
            same = (img == ref).all(axis=1)
            close = (np.abs(img - ref) <= 1e-9 * np.maximum(1.0, np.abs(ref))).all(axis=1)
            print("%-8s %-28s bit-equal pixels %.4f %%, within 1e-9 %.4f %%, max |delta| %.3e, mean rel. error %.2e"
                  % (name, label, 100 * same.mean(), 100 * close.mean(), np.abs(img - ref).max(),
                     abs(img.mean() - ref.mean()) / max(1e-300, abs(ref.mean()))))
            worst = max(worst, 1.0 - close.mean())
        if save:
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_%s.npz" % name), image=ref, width=w, height=h,
                                max_bounces=b, iterations=n, seed=seed)
    sys.exit(0 if worst < 0.02 else 1)


if __name__ == "__main__":
    main()
