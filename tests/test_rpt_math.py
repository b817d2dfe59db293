"""include/rpt_math.h (the fdlibm restatement both the oracle and the kernels evaluate) against the
host libm: every function within 1 ulp of glibc over millions of arguments, special values exact.
Also: the oracle built with glibc (liboracle_sysm.so) renders the same images up to the rare
branch flips a differing ulp causes."""
import ctypes as C
import math

import numpy as np

import rpt_amd
from rpt_amd import make_params


def ulp_diff(a, b):
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2 ** 63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2 ** 63) - ib[ib < 0]
    return np.abs(ia - ib)


def check(oracle, fn, x, ref, y=None, max_ulp=1, min_exact=0.8):
    got = oracle.math_eval(fn, x, y)
    finite = np.isfinite(ref)
    assert (np.isnan(got) == np.isnan(ref)).all()
    assert (got[~finite & ~np.isnan(ref)] == ref[~finite & ~np.isnan(ref)]).all()
    d = ulp_diff(got[finite], ref[finite])
    assert d.max() <= max_ulp, (fn, d.max(), x[finite][d.argmax()])
    assert (d == 0).mean() >= min_exact, (fn, (d == 0).mean())


def test_exp(oracle):
    rs = np.random.RandomState(1)
    x = np.concatenate([rs.uniform(-745, 709.7, 1000000), rs.uniform(-1, 1, 500000), -rs.exponential(5.0, 500000),
                        [0.0, -0.0, 1.0, -1.0, 709.78, 709.79, -745.2, np.inf, -np.inf, np.nan, 1e-300, -1e-10]])
    with np.errstate(all="ignore"):
        check(oracle, 0, x, np.exp(x))


def test_log(oracle):
    rs = np.random.RandomState(2)
    x = np.concatenate([rs.rand(1000000), np.exp(rs.uniform(-700, 700, 500000)), 1 + rs.uniform(-1e-5, 1e-5, 200000),
                        [0.0, 1.0, 2.0, 5e-324, 1e-310, np.inf, np.nan, -1.0]])
    with np.errstate(all="ignore"):
        check(oracle, 1, x, np.log(x))


def test_atan(oracle):
    rs = np.random.RandomState(3)
    x = np.concatenate([rs.uniform(-3, 3, 1000000), np.exp(rs.uniform(-40, 60, 500000)), -np.exp(rs.uniform(-40, 60, 100000)),
                        [0.0, -0.0, 1.0, 0.4375, 0.6875, 1.1875, 2.4375, np.inf, -np.inf, np.nan, 1e-300]])
    check(oracle, 2, x, np.arctan(x))


def test_sin_cos_on_the_contract_range(oracle):
    rs = np.random.RandomState(4)
    x = np.concatenate([rs.uniform(0, math.pi / 2, 1500000), rs.uniform(-3 * math.pi / 4, 3 * math.pi / 4, 500000) * 0.999,
                        [0.0, math.pi / 4, math.pi / 2, 1e-30, 0.78539816339744828, 1.5707963267948966]])
    check(oracle, 3, x, np.sin(x))
    check(oracle, 4, x, np.cos(x))
    assert np.isnan(oracle.math_eval(3, np.array([3.0, 100.0, np.inf, np.nan]))).all()  # out of contract -> NaN


def test_acos(oracle):
    rs = np.random.RandomState(5)
    x = np.concatenate([rs.uniform(-1, 1, 1500000), 1 - np.exp(rs.uniform(-40, 0, 200000)), [1.0, -1.0, 0.0, 0.5, -0.5, 1.5, np.nan]])
    with np.errstate(all="ignore"):
        check(oracle, 5, x, np.arccos(x))


def test_atan2(oracle):
    rs = np.random.RandomState(6)
    y = np.concatenate([rs.randn(1500000), [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.inf, 1.0, np.nan, 1e300, 1e-300]])
    x = np.concatenate([rs.randn(1500000), [1.0, 1.0, -1.0, -1.0, 0.0, 0.0, np.inf, -np.inf, 1.0, np.inf, 1.0, 1e-300, -1e300]])
    check(oracle, 6, x, np.arctan2(y, x), y=y)


def test_oracle_with_system_libm_renders_the_same(oracle):
    """Two builds of the oracle — fdlibm restatement vs the host's glibc — differ only where a
    1-ulp difference flips a branch downstream: almost all pixels agree to 1e-9, and the images
    agree statistically."""
    scene, cam, _ = rpt_amd.scenes.cornell()
    desc, keep = scene.lower()
    p = make_params(96, 54, 4, 8, seed=11)
    camc = cam.lower()
    outs = []
    for L in (oracle.lib(), oracle.sysm_lib()):
        h = C.c_void_p()
        assert L.oracle_scene_create(C.byref(desc), C.byref(h)) == 0
        out = np.empty((96 * 54, 3))
        assert L.oracle_render(h, C.byref(camc), C.byref(p), 0, out.ctypes.data_as(C.POINTER(C.c_double)), None) == 0
        L.oracle_scene_destroy(h)
        outs.append(out)
    a, b = outs
    close = (np.abs(a - b) <= 1e-9 * np.maximum(1.0, np.abs(a))).all(axis=1)
    assert close.mean() > 0.98
    assert abs(a.mean() - b.mean()) / a.mean() < 2e-3


def test_against_multiprecision_reference(oracle):
    """Independent of any libm: include/rpt_math.h against mpmath at 60 digits.  Every function is within 1 ulp of
    the TRUE value (the fdlibm bound) on arguments spread over the ranges the path tracer uses — so "GPU ≡ oracle"
    on these six functions means "both evaluate an accurate exp/ln/atan/sin/cos/acos/atan2", not merely "both
    evaluate the same thing"."""
    import mpmath
    mpmath.mp.dps = 60
    rs = np.random.RandomState(11)
    n = 4000

    def ulps_off(got, exact_mp):
        out = np.empty(len(got))
        for i, (g, e) in enumerate(zip(got, exact_mp)):
            r = float(e)  # correctly rounded double
            if r == 0.0 or not math.isfinite(r):
                out[i] = 0.0 if g == r else np.inf
                continue
            ulp = math.ulp(r)
            out[i] = abs(mpmath.mpf(float(g)) - e) / ulp
        return out

    cases = [
        (0, np.concatenate([rs.uniform(-700, 700, n), rs.uniform(-2, 2, n), -rs.exponential(3.0, n)]), None, mpmath.exp),
        (1, np.concatenate([rs.uniform(0, 1, n), np.exp(rs.uniform(-300, 300, n)), 1.0 + rs.uniform(-1e-3, 1e-3, n)]), None, mpmath.log),
        (2, np.concatenate([np.exp(rs.uniform(-20, 20, n)), rs.uniform(-3, 3, n)]), None, mpmath.atan),
        (3, rs.uniform(0, np.pi / 2, 2 * n), None, mpmath.sin),
        (4, rs.uniform(0, np.pi / 2, 2 * n), None, mpmath.cos),
        (5, np.concatenate([rs.uniform(-1, 1, n), 1.0 - np.exp(rs.uniform(-30, 0, n)), -1.0 + np.exp(rs.uniform(-30, 0, n))]), None, mpmath.acos),
    ]
    for fn, x, y, ref in cases:
        got = oracle.math_eval(fn, x, y)
        off = ulps_off(got, [ref(mpmath.mpf(float(v))) for v in x])
        assert off.max() < 1.0, (fn, off.max(), x[off.argmax()])
        assert (off <= 0.5).mean() > 0.8, (fn, (off <= 0.5).mean())   # mostly correctly rounded
    yy, xx = rs.randn(2 * n), rs.randn(2 * n)
    got = oracle.math_eval(6, xx, yy)
    off = ulps_off(got, [mpmath.atan2(mpmath.mpf(float(a)), mpmath.mpf(float(b))) for a, b in zip(yy, xx)])
    assert off.max() < 1.5, off.max()   # atan2 = atan of a rounded quotient + a rounded pi fold: fdlibm's bound is 2 ulp
