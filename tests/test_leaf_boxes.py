"""The conservative leaf boxes in front of Triangle::intersect (rpt_amd/csrc/host_scene.cpp fill_leaf_boxes,
kernels/shapes.inc leaf_box_pass / boxray_make) restated in numpy, and their one obligation checked on millions of
(ray, triangle) pairs: **whenever the exact test of mesh.rs:49-82 accepts, the filter passes** — for the widest window
and for windows that barely contain the hit — including rays that graze edges and vertices, rays almost parallel to
an axis, triangles at the corners of the grid, slivers, and origins far from the mesh.  (That the DEVICE code is this
filter is what the bit-exact GPU parity tests show: a filter that dropped an accepted hit would change an image.)"""
import numpy as np

from rpt_amd import scenes

GRID = 65529.0   # plus two steps of padding on either side of the bounds
BOX_PAR = 1e30   # slab "slope" of an axis with d == 0 (shapes.inc)


def quantise(tris, lo, hi):
    """fill_leaf_boxes: (n, 6) uint grid boxes of (n, 9) triangles in the tree bounds [lo, hi]"""
    ext = hi - lo
    scale = np.where((ext > 0) & np.isfinite(ext), ext / GRID, 1.0)
    lo = lo - 2.0 * scale
    v = tris.reshape(-1, 3, 3)
    bmin, bmax = v.min(axis=1), v.max(axis=1)
    a = np.clip(np.floor((bmin - lo) / scale) - 1.0, 0.0, 65535.0)
    c = np.clip(np.ceil((bmax - lo) / scale) + 1.0, 0.0, 65535.0)
    d0, d1 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    d00, d01, d11 = (d0 * d0).sum(1), (d0 * d1).sum(1), (d1 * d1).sum(1)
    denom = d00 * d11 - d01 * d01
    full = ~(denom > 1e-10 * (d00 * d11)) | ~np.isfinite(denom)
    a[full], c[full] = 0.0, 65535.0
    # the decoded box contains the true box with a margin of (almost) one step on every side — also on the bounds' faces
    nf = ~full
    assert (lo + a[nf] * scale <= bmin[nf] - 0.999 * scale).all() and (lo + c[nf] * scale >= bmax[nf] + 0.999 * scale).all()
    return np.concatenate([a, c], axis=1), scale, lo, full


def outward(x, sign):
    """box_window: one f32 ulp and a bit more, outwards; NaN stays NaN"""
    with np.errstate(all="ignore"):
        x = x.astype(np.float32)
        return (x + np.float32(sign) * (np.abs(x) * np.float32(1.2e-7) + np.float32(1e-30))).astype(np.float32)


def box_pass(q, scale, lo, o, d, t_lo, t_hi, far=1e12):
    """boxray_make + box_window + leaf_box_pass for pairs (q[i], ray i); returns (pass, filter_on).  The device evaluates
    the slabs in f32 with fmaf, counted from t0 = max(root slab entry, 0); numpy has no fmaf: q * a is exact in f64
    (16 x 24 bits), so f64 add + one rounding to f32 differs from it only in rare double-rounding ties.  q arrives as
    (lo, hi) grid coordinates; the stored form — centre and half-extent — is derived here as the host derives it."""
    with np.errstate(all="ignore"):
        z = d == 0                      # axes the ray does not move along: "is the origin inside the box on that axis"
        r = 1.0 / d
        g = (o - lo) / scale
        # t0: where the ray enters the bounds the grid was made for (grid coordinates 2 .. 65531), or 0 from inside
        f0, f1 = (lo + 2.0 * scale - o) * r, (lo + (GRID + 2.0) * scale - o) * r
        b_min = np.fmax(np.fmax(np.fmin(f0, f1)[:, 0], np.fmin(f0, f1)[:, 1]), np.fmin(f0, f1)[:, 2])
        t0 = np.fmax(b_min, 0.0)
        a = np.where(z, BOX_PAR, scale * r)
        u = (lo - o) * r
        b = np.where(z, -g * BOX_PAR, u - t0[:, None])
        ok = np.where(z, np.abs(g) < 1e6, (np.abs(a) > 1e-30) & (np.abs(a) < 1e30) & (np.abs(b) < 1e6 * np.abs(a))
                      & (np.abs(u) < far * np.abs(a)))
        on = ~z.all(axis=1) & ok.all(axis=1) & (np.abs(t0) < 1e300)
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        # stored as centre and half-extent (quantise_box): c = floor((lo + hi) / 2), h = hi - c; the slab ends of an axis
        # are t(c) -+ h |a|: one fma for the centre, one (packed) fma for both ends, whatever the direction's sign
        qc = np.floor((q[:, 0:3] + q[:, 3:6]) / 2.0)
        qh = q[:, 3:6] - qc
        assert (qc - qh <= q[:, 0:3]).all() and (qc - qh >= q[:, 0:3] - 1.0).all() and qh.max() <= 32768.0
        tc = (qc * a32.astype(np.float64) + b32.astype(np.float64)).astype(np.float32)
        h32 = np.abs(a32).astype(np.float64)
        near = (tc.astype(np.float64) - qh * h32).astype(np.float32)
        far = (tc.astype(np.float64) + qh * h32).astype(np.float32)
        tl = np.fmax(np.fmax(near[:, 0], near[:, 1]), near[:, 2])
        th = np.fmin(np.fmin(far[:, 0], far[:, 1]), far[:, 2])
        wl = outward(np.broadcast_to(np.asarray(t_lo, dtype=np.float64), t0.shape) - t0, -1.0)
        wh = outward(np.broadcast_to(np.asarray(t_hi, dtype=np.float64), t0.shape) - t0, +1.0)
        ok = (tl <= th) & ~(th < wl) & ~(tl > wh)   # a NaN bound (t_min below a 0/0 split) rejects nothing
    return ok | ~on, on


def exact_hit(tris, o, d):
    """Triangle::intersect (mesh.rs:49-82) with t_min = 1e-12 and record.time = +inf: (accepted, time)"""
    with np.errstate(all="ignore"):
        v1, v2, v3 = tris[:, 0:3], tris[:, 3:6], tris[:, 6:9]
        d0, d1 = v2 - v1, v3 - v1
        n = np.cross(d0, d1)
        n = n / np.sqrt((n * n).sum(1, keepdims=True))
        cosine = (n * d).sum(1)
        time = (n * (v1 - o)).sum(1) / cosine
        d2 = (o + time[:, None] * d) - v1
        d00, d01, d11 = (d0 * d0).sum(1), (d0 * d1).sum(1), (d1 * d1).sum(1)
        d20, d21 = (d2 * d0).sum(1), (d2 * d1).sum(1)
        denom = d00 * d11 - d01 * d01
        v = (d11 * d20 - d01 * d21) / denom
        w = (d00 * d21 - d01 * d20) / denom
        u = 1.0 - v - w
        acc = ~(np.abs(cosine) < 1e-8) & ~((time < 1e-12)) & (u >= 0) & (v >= 0) & (w >= 0)
    return acc, time


def check(tris, o, d, lo, hi, what):
    q, scale, lo, full = quantise(tris, lo, hi)
    acc, time = exact_hit(tris, o, d)
    assert acc.sum() > 0.1 * len(acc), (what, acc.mean())  # rays aimed at a vertex hit it ~1 time in 5
    for t_lo, t_hi in ((1e-12, np.inf), (time, time), (np.nextafter(time, -np.inf), np.nextafter(time, np.inf)),
                       (np.nan, np.nextafter(time, np.inf))):
        ok, on = box_pass(q, scale, lo, o, d, t_lo, t_hi)
        bad = acc & ~ok
        assert not bad.any(), (what, int(bad.sum()), np.flatnonzero(bad)[:5])
    return on.mean(), full.mean()


def aimed_rays(rs, tris, bary, dist_scale=1.0):
    """rays from random origins through the point with barycentrics `bary` of each triangle"""
    v = tris.reshape(-1, 3, 3)
    p = (bary[:, :, None] * v).sum(axis=1)
    o = p + rs.randn(len(tris), 3) * dist_scale
    d = p - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def test_filter_never_rejects_an_accepted_hit_on_the_bench_mesh():
    rows = scenes.knot_mesh(nu=392, nv=32)            # 25k triangles of the C3 stand-in's shape
    tris = rows[:, :9]
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    rs = np.random.RandomState(1)
    rep = np.repeat(tris, 40, axis=0)                 # 10^6 pairs per case
    for what, bary in (("interior", rs.dirichlet((1, 1, 1), len(rep))),
                       ("edges", np.stack([rs.rand(len(rep)), 1 - rs.rand(len(rep)) * 0 - 0, np.zeros(len(rep))], 1)),
                       ("vertices", np.eye(3)[rs.randint(0, 3, len(rep))])):
        if what == "edges":
            a = rs.rand(len(rep))
            bary = np.stack([a, 1.0 - a, np.zeros(len(rep))], axis=1)[:, rs.permutation(3)]
        for dist in (0.05, 1.0, 50.0):
            o, d = aimed_rays(rs, rep, bary, dist)
            on, full = check(rep, o, d, lo, hi, (what, dist))
            assert on > 0.99 and full == 0.0


def test_filter_with_axis_parallel_rays_far_origins_and_corner_triangles():
    rs = np.random.RandomState(2)
    n = 400000
    lo, hi = np.array([-3.0, 0.5, 10.0]), np.array([5.0, 0.75, 4000.0])      # anisotropic bounds, off-centre
    c = lo + rs.rand(n, 3) * (hi - lo)
    tris = (c[:, None, :] + rs.randn(n, 3, 3) * (hi - lo) * 10.0 ** rs.uniform(-4, -1, (n, 1, 1))).reshape(n, 9)
    tris = np.clip(tris.reshape(n, 3, 3), lo, hi).reshape(n, 9)                # some flattened onto the bounds' faces
    bary = rs.dirichlet((1, 1, 1), n)
    o, d = aimed_rays(rs, tris, bary, 3.0)
    check(tris, o, d, lo, hi, "anisotropic")
    # nearly axis-parallel directions: huge slab parameters on two axes
    p = (bary[:, :, None] * tris.reshape(n, 3, 3)).sum(1)
    axis = rs.randint(0, 3, n)
    d = rs.randn(n, 3) * 1e-9
    d[np.arange(n), axis] = rs.choice([-1.0, 1.0], n)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = p - d * rs.uniform(0.1, 100.0, (n, 1))
    check(tris, o, d, lo, hi, "axis-parallel")
    # EXACTLY axis-parallel directions (one or two components zero: shadow rays towards an axis-aligned directional
    # light): the filter stays on, and passes whenever the exact test accepts
    for nz in (1, 2):
        d = rs.randn(n, 3)
        for k in range(nz):
            d[np.arange(n), (axis + k) % 3] = 0.0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = p - d * rs.uniform(0.1, 100.0, (n, 1))
        o = np.where(d == 0, p, o)         # o + t d == p is then exact on the zero axes
        on, _ = check(tris, o, d, lo, hi, "%d zero components" % nz)
        assert on > 0.95
        # and it does filter: against OTHER triangles (shifted by a tenth of the extent) almost nothing passes
        q, scale, lo2, full = quantise(np.roll(tris, n // 2, axis=0), lo, hi)
        ok, _ = box_pass(q, scale, lo2, o, d, 1e-12, np.inf)
        assert ok.mean() < 0.2, ok.mean()
    # origins up to 10^10 extents away: the slab parameters are counted from where the ray enters the bounds, so the
    # filter stays on until the f64 cancellation in "parameter - entry" would matter (10^12 grid steps), then is off
    o, d = aimed_rays(rs, tris, bary, 1.0)
    far = 10.0 ** rs.uniform(0, 10, (n, 1)) * np.linalg.norm(hi - lo)
    o = p - d * far
    on, _ = check(tris, o, d, lo, hi, "far origins")
    assert 0.2 < on < 0.95, on   # (the thin axis of these bounds reaches 10^12 of ITS steps first)


def test_slivers_are_never_filtered():
    rs = np.random.RandomState(3)
    n = 100000
    a = rs.randn(n, 3)
    e = rs.randn(n, 3)
    tris = np.concatenate([a, a + e, a + e * rs.uniform(0.2, 0.8, (n, 1)) + rs.randn(n, 3) * 1e-7], axis=1)   # sin < 1e-5
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    q, scale, _, full = quantise(tris, lo, hi)
    assert full.mean() > 0.99 and (q[full, 0:3] == 0).all() and (q[full, 3:6] == 65535).all()
    degenerate = np.concatenate([a, a, a + e], axis=1)     # zero area: denom == 0
    _, _, _, full = quantise(degenerate, lo, hi)
    assert full.all()


# ---- the same filter in front of whole objects: placed spheres and cubes -------------------------------------------
# (group-tree children since round 2, top-level objects of flat scenes since round 3: host_scene.cpp fill_object_boxes)
def random_placements(rs, n, smin, smax):
    """M = T R S (rotation by a random orthogonal matrix, scales in [smin, smax]): (M 3x3, translation, inverse 3x3)"""
    qm, _ = np.linalg.qr(rs.randn(n, 3, 3))
    sc = np.exp(rs.uniform(np.log(smin), np.log(smax), (n, 3)))
    lin = qm * sc[:, None, :]
    tr = rs.uniform(-3.0, 3.0, (n, 3))
    return lin, tr, np.linalg.inv(lin)


def world_boxes(lin, tr, half):
    """Transformed::bounding_box (shape.rs:153-176): the box of the eight transformed corners of [-half, half]^3"""
    corners = np.array([[sx, sy, sz] for sx in (-half, half) for sy in (-half, half) for sz in (-half, half)])
    w = np.einsum("nij,cj->nci", lin, corners) + tr[:, None, :]
    return w.min(axis=1), w.max(axis=1)


def to_object_space(inv, tr, o, d):
    lo = np.einsum("nij,nj->ni", inv, o - tr)
    ld = np.einsum("nij,nj->ni", inv, d)
    return lo, ld


def sphere_hit(lo, ld):
    """Sphere::intersect (sphere.rs:13-45) with t_min = 1e-12 and record.time = +inf: (accepted, time)"""
    with np.errstate(all="ignore"):
        a = (ld * ld).sum(1)
        b = 2.0 * (lo * ld).sum(1)
        c = (lo * lo).sum(1) - 1.0
        disc = b * b - 4.0 * a * c
        sq = np.sqrt(disc)
        t1, t2 = (-b - sq) / (2.0 * a), (-b + sq) / (2.0 * a)
        t = np.where(t1 >= 1e-12, t1, t2)
        acc = (disc >= 0) & (t >= 1e-12)
    return acc, t


def cube_hit(lo, ld):
    """Cube::intersect (cube.rs:20-72): slabs of [-0.5, 0.5]^3, entry time if it is >= t_min, else the exit time"""
    with np.errstate(all="ignore"):
        t0, t1 = (-0.5 - lo) / ld, (0.5 - lo) / ld
        near, far = np.fmin(t0, t1), np.fmax(t0, t1)
        tn, tf = near.max(axis=1), far.min(axis=1)
        t = np.where(tn >= 1e-12, tn, tf)
        acc = (tn <= tf) & (t >= 1e-12)
    return acc, t


def quantise_boxes(bmin, bmax, lo, hi):
    ext = hi - lo
    scale = np.where((ext > 0) & np.isfinite(ext), ext / GRID, 1.0)
    lo = lo - 2.0 * scale
    a = np.clip(np.floor((bmin - lo) / scale) - 1.0, 0.0, 65535.0)
    c = np.clip(np.ceil((bmax - lo) / scale) + 1.0, 0.0, 65535.0)
    return np.concatenate([a, c], axis=1), scale, lo


def test_object_filter_never_rejects_a_hit_on_a_placed_sphere_or_cube():
    rs = np.random.RandomState(5)
    n = 400000
    for what, half, hit in (("sphere", 1.0, sphere_hit), ("cube", 0.5, cube_hit)):
        lin, tr, inv = random_placements(rs, n, 0.05, 2.0)          # condition numbers up to 40, semi-axes >= 0.05
        bmin, bmax = world_boxes(lin, tr, half)
        lo, hi = bmin.min(0), bmax.max(0)                           # the grid spans all of them (about 10 units)
        q, scale, glo = quantise_boxes(bmin, bmax, lo, hi)
        assert (0.05 >= 64.0 * scale).all()                         # no sphere is "too small" (quadric_too_small)
        # rays aimed at points of the object's surface neighbourhood (so that about half of them graze or miss), from
        # 0.1 to 3000 units away (1e7 steps are 1500 units: beyond that the filter must be off, see below)
        p_obj = rs.randn(n, 3)
        p_obj /= np.linalg.norm(p_obj, axis=1, keepdims=True)
        p_obj *= half * rs.uniform(0.9, 1.15, (n, 1))
        if what == "cube":
            p_obj = rs.uniform(-0.9, 0.9, (n, 3))
        p = np.einsum("nij,nj->ni", lin, p_obj) + tr
        dist = 10.0 ** rs.uniform(-1, 3.5, (n, 1))
        d = rs.randn(n, 3)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = p - d * dist
        lo_, ld_ = to_object_space(inv, tr, o, d)
        acc, t = hit(lo_, ld_)
        assert 0.2 < acc.mean() < 0.95, (what, acc.mean())
        for t_lo, t_hi in ((1e-12, np.inf), (t, t), (np.nextafter(t, -np.inf), np.nextafter(t, np.inf))):
            ok, on = box_pass(q, scale, glo, o, d, t_lo, t_hi, far=1e7)
            bad = acc & ~ok
            assert not bad.any(), (what, int(bad.sum()), np.flatnonzero(bad)[:5])
        assert 0.5 < on.mean() < 0.999, on.mean()                   # on nearby, off from far away
        # and it filters: against the box of ANOTHER object almost every ray is rejected
        ok, _ = box_pass(np.roll(q, n // 2, axis=0), scale, glo, o, d, 1e-12, np.inf, far=1e7)
        assert ok[on].mean() < 0.25, ok[on].mean()


def test_why_far_origins_switch_the_filter_off_for_spheres():
    # from 10^9 radii away Sphere::intersect accepts lines that miss the sphere by many radii (b^2 - 4ac cancels):
    # the reference's answer is the contract, the box cannot contain such "hits", hence the 1e7-step limit
    rs = np.random.RandomState(6)
    n = 200000
    r = 2e-3
    o = np.array([0.0, 0.0, 6.0e7]) + rs.randn(n, 3)
    tgt = rs.randn(n, 3) * 0.5                                       # aimed up to hundreds of radii off the centre
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    acc, t = sphere_hit(o / r, d / r)
    with np.errstate(all="ignore"):
        miss = np.linalg.norm(np.cross(-o, d), axis=1)               # distance of the line from the centre
    wrong = acc & (miss > 10 * r)
    assert wrong.sum() > 100, int(wrong.sum())
    q, scale, glo = quantise_boxes(np.full((n, 3), -r), np.full((n, 3), r), np.array([-3.0] * 3), np.array([3.0] * 3))
    ok, on = box_pass(q, scale, glo, o, d, 1e-12, np.inf, far=1e7)
    assert not on.any() and ok.all()
