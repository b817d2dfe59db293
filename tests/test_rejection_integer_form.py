"""Triangle::sample's rejection test (reference src/shape/mesh.rs:84-98: `while u + v > 1.0`) is decided on the draws'
integers in the device code (rpt_amd/csrc/kernels/sampling.inc sample_tri): with u = A / 2^53 and v = B / 2^53
(rand 0.8 Standard f64: the top 53 bits of a u64), `u + v > 1.0` in IEEE double arithmetic  <=>  A + B >= 2^53 + 2.
Checked here against numpy's float64 — the oracle keeps the floating-point form, so the GPU parity tests check it too —
on random pairs and exhaustively around the boundary, the one rounding tie (A + B = 2^53 + 1 rounds to 2^53) included."""
import numpy as np

TWO53 = 1 << 53


def f64_form(a, b):
    u = a.astype(np.float64) * (1.0 / 9007199254740992.0)
    v = b.astype(np.float64) * (1.0 / 9007199254740992.0)
    return (u + v) > 1.0


def int_form(a, b):
    return (a + b) >= np.uint64(TWO53 + 2)


def test_random_pairs():
    rng = np.random.default_rng(7)
    a = rng.integers(0, TWO53, size=4_000_000, dtype=np.uint64)
    b = rng.integers(0, TWO53, size=4_000_000, dtype=np.uint64)
    assert (f64_form(a, b) == int_form(a, b)).all()
    assert 0.49 < f64_form(a, b).mean() < 0.51  # the loop's rejection probability: one half


def test_every_sum_near_the_boundary():
    rng = np.random.default_rng(8)
    for d in range(-8, 12):
        total = TWO53 + d
        a = rng.integers(max(0, total - (TWO53 - 1)), min(total, TWO53 - 1) + 1, size=100_000, dtype=np.uint64)
        b = np.uint64(total) - a
        assert (b < TWO53).all()
        f, i = f64_form(a, b), int_form(a, b)
        assert (f == i).all(), d
        assert f.all() == (d >= 2) and f.any() == (d >= 2), d  # d = 1 is the tie: the sum rounds to exactly 1.0
    # the extremes of the draws themselves
    e = np.array([0, 1, TWO53 - 1, TWO53 // 2, TWO53 // 2 + 1], dtype=np.uint64)
    a, b = np.meshgrid(e, e)
    assert (f64_form(a.ravel(), b.ravel()) == int_form(a.ravel(), b.ravel())).all()
