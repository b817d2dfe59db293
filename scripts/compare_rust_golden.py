#!/usr/bin/env python
"""Pins the CPU oracle to the REFERENCE: compares frames dumped by the patched reference
(`cargo run --release --features philox --example dump_golden -- DIR`, see rust/rpt.patch and
rust/rpt_additions/dump_golden.rs) with the oracle's render of the same configuration and seed.

    python scripts/compare_rust_golden.py DIR [--rpt-root CHECKOUT] [--save-fixtures]

Expected: bit-equal on almost every pixel; the oracle evaluates exp/ln/atan/sin_cos/acos/atan2 with the fdlibm
restatement of include/rpt_math.h while Rust calls the platform libm (<= 1 ulp apart on ~10 % of arguments), so a
small fraction of pixels may differ in the last bits or by one re-rolled branch.  Ends with ONE line,
`pin_oracle: PASS` or `pin_oracle: FAIL (...)`, and the exit code says the same.  With --save-fixtures the reference's
frames are stored as tests/golden/rust_<name>.npz (the teapot scene's fixture carries the parsed triangles of the crate's
examples/teapot.obj, so the tests need no asset); from then on tests/test_rust_golden.py holds the oracle AND — in the GPU
suite — the HIP path to them, and the "parity unpinned" notes (oracle/oracle.cpp's header) can go.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi as O  # noqa: E402
from rpt_amd import make_params  # noqa: E402
from rpt_amd import golden_scenes  # noqa: E402

# the share of pixels that must agree with the reference within 1e-9 (relative, per channel).  Not 100 %: Rust's f64::exp /
# ln / atan / sin_cos are the platform libm's, the oracle's are fdlibm restated (<= 1 ulp apart on a tenth of the
# arguments), and one differing ulp can re-roll a rejection loop or a lobe choice of that sample.
MIN_CLOSE = 0.98


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    d = args[0]
    save = "--save-fixtures" in sys.argv
    rpt_root = None
    if "--rpt-root" in sys.argv:
        rpt_root = sys.argv[sys.argv.index("--rpt-root") + 1]
        args = [a for a in args if a != rpt_root]
        d = args[0]
    failures, compared = [], 0
    for name in golden_scenes.NAMES:
        txt = os.path.join(d, name + ".txt")
        if not os.path.exists(txt):
            print("%-8s no dump in %s" % (name, d))
            if name != "teapot":  # (the teapot scene needs the crate's asset: optional)
                failures.append("%s: no dump" % name)
            continue
        kv = dict(line.split() for line in open(txt))
        w, h, b, n, seed = (int(kv[k]) for k in ("width", "height", "max_bounces", "iterations", "seed"))
        ref = np.fromfile(os.path.join(d, name + ".f64"), dtype="<f8").reshape(h * w, 3)
        extra = {}
        if name == "teapot":
            obj = os.path.join(rpt_root or ".", "examples", "teapot.obj")
            if not os.path.exists(obj):
                failures.append("teapot: %s not found (pass --rpt-root)" % obj)
                continue
            from rpt_amd import io as rio
            extra["triangles"] = np.ascontiguousarray(rio.load_obj(obj).triangles, dtype=np.float64)
        scene, cam = golden_scenes.build(name, **extra)
        p = make_params(w, h, b, n, seed=seed)
        compared += 1
        for lib, label, decides in ((O.lib(), "oracle (fdlibm restatement)", True),
                                    (O._load(os.path.join(ROOT, "oracle", "liboracle_sysm.so")), "oracle (system libm)", False)):
            img = O.OracleScene(scene, lib).render(cam, p, threads=0)
            same = (img == ref).all(axis=1)
            close = (np.abs(img - ref) <= 1e-9 * np.maximum(1.0, np.abs(ref))).all(axis=1)
            print("%-8s %-28s bit-equal pixels %.4f %%, within 1e-9 %.4f %%, max |delta| %.3e, mean rel. error %.2e"
                  % (name, label, 100 * same.mean(), 100 * close.mean(), np.abs(img - ref).max(),
                     abs(img.mean() - ref.mean()) / max(1e-300, abs(ref.mean()))))
            if decides and close.mean() < MIN_CLOSE:
                failures.append("%s: only %.2f %% of pixels within 1e-9 of the reference" % (name, 100 * close.mean()))
        if save:
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rust_%s.npz" % name), image=ref, width=w, height=h,
                                max_bounces=b, iterations=n, seed=seed, **extra)
    if compared == 0:
        failures.append("nothing compared")
    if failures:
        print("pin_oracle: FAIL (%s)" % "; ".join(failures))
        sys.exit(1)
    print("pin_oracle: PASS — %d scene(s): the oracle reproduces the reference's frames%s"
          % (compared, "; fixtures written to tests/golden/rust_*.npz — commit them" if save else ""))
    sys.exit(0)


if __name__ == "__main__":
    main()
