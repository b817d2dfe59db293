"""A second, independent evaluation of the GEOMETRIC half of the path — the discipline of test_independent_shading.py
carried over to everything the GPU-vs-oracle parity tests see through one shared reading only.

`oracle/oracle.cpp` and `rpt_amd/csrc/kernels/{shapes,sampling,light,paths}.inc` are twins.  This file restates, in
numpy and in its own decomposition (vectorised, masks instead of early returns), written from the Rust text alone —

  Sphere::intersect      src/shape/sphere.rs:13-45        Sphere::sample       src/shape/sphere.rs:52-64
  Plane::intersect       src/shape/plane.rs:17-32         Cube::sample         src/shape/cube.rs:74-87
  Cube::intersect        src/shape/cube.rs:20-72          Triangle::sample     src/shape/mesh.rs:84-98
  Triangle::intersect    src/shape/mesh.rs:49-82          KdTree::sample       src/kdtree.rs:138-143
  BoundingBox::intersect src/kdtree.rs:53-69              Transformed::sample  src/shape.rs:139-151
  KdTree::intersect's root test  src/kdtree.rs:129-135    Camera::cast_ray     src/camera.rs:64-81 (+ get_color's pixel
  Ray::apply_transform, Transformed::intersect            mapping and jitter, src/renderer.rs:131-139)
                         src/shape.rs:64-71, 128-137      Hdri::get_color      src/environment.rs:25-52
  the firefly clamp's fold  src/renderer.rs:152-167

(with rand 0.8.3's `gen::<f64>`, `gen_range` for floats, `Uniform::from(0..n)` for integers and rand_distr 0.4's
`UnitDisc` from their published sources) — and holds each against the oracle on >= 10^4 random cases: the same accept /
reject decisions, the same number of random draws, values to a few ulp.  NOT from oracle.cpp; the two sides share no
code, and the matrices of a Transformed shape are numpy's (LU) here and rpt_amd/glm.py's (cofactors) there.

No GPU, no reference checkout at run time (the line numbers are citations)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_ffi as O  # noqa: E402
from rpt_amd import Camera, Environment, KdTree, Mesh, cube, make_params, plane, scenes, sphere  # noqa: E402
from rpt_amd.shape import Triangle  # noqa: E402
from test_independent_shading import Stream, ulps, unit  # noqa: E402

INF = math.inf


def d3(a, b):
    """nalgebra's 3-vector dot: the products summed left to right"""
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def norm3(a):
    return np.sqrt(d3(a, a))


def normalize(a):
    return a / norm3(a)[..., None]


def random_rays(rs, n, reach=3.0):
    o = rs.uniform(-reach, reach, (n, 3))
    d = rs.normal(size=(n, 3))
    aim = rs.rand(n) < 0.7  # most of them towards the unit shapes around the origin, so that hits are common
    d[aim] = rs.uniform(-0.6, 0.6, (aim.sum(), 3)) - o[aim]
    return o, unit(d) * rs.uniform(0.5, 2.0, (n, 1))  # (directions are not normalised in object space)


# =============================================================================================== intersections
def sphere_intersect(o, d, t_min, time):
    """Sphere::intersect (sphere.rs:13-45) -> (hit, t, normal)"""
    a, b, c = d3(d, d), d3(d, o), d3(o, o) - 1.0
    disc = b * b - a * c
    with np.errstate(all="ignore"):
        root = np.sqrt(disc)
        t_minus, t_plus = (-b - root) / a, (-b + root) / a
    use_plus = t_minus < t_min
    t = np.where(use_plus, t_plus, t_minus)
    miss = np.signbit(disc) | (use_plus & (t_plus < t_min))     # :18-20, :27-29
    hit = ~miss & (t < time)                                    # :36
    return hit, t, normalize(o + t[:, None] * d)


def plane_intersect(nrm, value, o, d, t_min, time):
    """Plane::intersect (plane.rs:17-32)"""
    nrm = np.broadcast_to(np.asarray(nrm, float), o.shape)
    cosine = d3(nrm, d)
    with np.errstate(all="ignore"):
        t = (value - d3(nrm, o)) / cosine
    hit = ~(np.abs(cosine) < 1e-8) & (t >= t_min) & (t < time)
    # -normal.normalize() * cosine.signum(): f64::signum is 1.0 for +0.0 and -1.0 for -0.0
    sgn = np.where(np.signbit(cosine), -1.0, 1.0)
    return hit, t, -normalize(nrm) * sgn[:, None]


def cube_intersect(o, d, t_min, time):
    """Cube::intersect (cube.rs:20-72)"""
    n = len(o)
    with np.errstate(all="ignore"):
        lo, hi = (-0.5 - o) / d, (0.5 - o) / d                   # compute_interval, per dimension
    swap = lo > hi
    near, far = np.where(swap, hi, lo), np.where(swap, lo, hi)
    near_sign, far_sign = np.where(swap, 1.0, -1.0), np.where(swap, -1.0, 1.0)
    x1, y1, z1 = near.T
    x2, y2, z2 = far.T
    s_ax = np.where((x1 > y1) & (x1 > z1), 0, np.where(y1 > z1, 1, 2))      # :37-45
    e_ax = np.where((x2 < y2) & (x2 < z2), 0, np.where(y2 < z2, 1, 2))      # :46-54
    rows = np.arange(n)
    start, end = near[rows, s_ax], far[rows, e_ax]
    sn, en = np.zeros((n, 3)), np.zeros((n, 3))
    sn[rows, s_ax] = near_sign[rows, s_ax]
    en[rows, e_ax] = far_sign[rows, e_ax]
    miss = (start > end) | (end < t_min)                         # :55-57
    inside = start < t_min                                       # :58
    t = np.where(inside, end, start)
    return ~miss & (t < time), t, np.where(inside[:, None], en, sn)


def bbox_intersect(box, o, d):
    """BoundingBox::intersect (kdtree.rs:53-69); f64::min / max ignore a NaN operand, like np.fmin / np.fmax"""
    with np.errstate(all="ignore"):
        a, b = (box[:3] - o) / d, (box[3:] - o) / d
    lo, hi = np.fmin(a, b), np.fmax(a, b)
    return np.fmax(np.fmax(lo[:, 0], lo[:, 1]), lo[:, 2]), np.fmin(np.fmin(hi[:, 0], hi[:, 1]), hi[:, 2])


def triangle_intersect(tri, o, d, t_min, time):
    """Triangle::intersect (mesh.rs:49-82); tri = (v1, v2, v3, n1, n2, n3)"""
    v1, v2, v3, n1, n2, n3 = (np.asarray(x, float) for x in tri)
    d0, d1 = v2 - v1, v3 - v1
    pn = np.cross(d0, d1)
    pn = pn / math.sqrt((pn[0] * pn[0] + pn[1] * pn[1]) + pn[2] * pn[2])
    pn_b = np.broadcast_to(pn, o.shape)
    cosine = d3(pn_b, d)
    with np.errstate(all="ignore"):
        t = d3(pn_b, v1 - o) / cosine
        p2 = (o + t[:, None] * d) - v1
        d00, d01, d11 = float(d3(d0, d0)), float(d3(d0, d1)), float(d3(d1, d1))
        d20, d21 = d3(p2, np.broadcast_to(d0, o.shape)), d3(p2, np.broadcast_to(d1, o.shape))
        denom = d00 * d11 - d01 * d01
        v = (d11 * d20 - d01 * d21) / denom
        w = (d00 * d21 - d01 * d20) / denom
        u = 1.0 - v - w
        nrm = normalize(u[:, None] * n1 + v[:, None] * n2 + w[:, None] * n3)
    plane_ok = ~(np.abs(cosine) < 1e-8) & ~((t < t_min) | (t >= time))      # :53-60
    return plane_ok & (u >= 0.0) & (v >= 0.0) & (w >= 0.0), t, nrm, (u, v, w)


def check_intersections(name, shape, mine, n=12000, seed=1):
    rs = np.random.RandomState(seed)
    o, d = random_rays(rs, n)
    t_min = np.where(rs.rand(n) < 0.2, rs.uniform(0.0, 3.0, n), 1e-12)
    time = np.where(rs.rand(n) < 0.3, rs.uniform(0.5, 6.0, n), INF)
    hit, t, nrm = mine(o, d, t_min, time)
    hits = 0
    for i in range(n):
        h0, t0, n0 = O.shape_intersect(shape, o[i], d[i], t_min[i], time[i])
        assert h0 == bool(hit[i]), (name, i, h0, t[i], t0)
        if h0:
            hits += 1
            assert ulps(t[i], t0) <= 4, (name, i, t[i], t0)
            assert np.abs(nrm[i] - n0).max() <= 4e-15 * max(1.0, abs(t0)), (name, i, nrm[i], n0)
        else:
            assert t0 == time[i]  # a miss leaves the record alone
    assert 0.15 * n < hits < 0.95 * n, (name, hits)
    return hits


def test_sphere_plane_cube_intersect_agree_with_an_independent_restatement():
    check_intersections("sphere", sphere(), sphere_intersect, seed=11)
    check_intersections("cube", cube(), cube_intersect, seed=12)
    for k, (nrm, value) in enumerate([((0.0, 1.0, 0.0), -0.3), ((0.3, -2.0, 0.7), 0.4), ((1e-3, 0.0, -5.0), 1.5)]):
        check_intersections("plane%d" % k, plane(nrm, value),
                            lambda o, d, a, b, nrm=nrm, value=value: plane_intersect(nrm, value, o, d, a, b), n=6000, seed=13 + k)


def test_axis_parallel_and_boundary_rays_agree():
    """zero direction components (the divisions give +-inf or 0/0 = NaN, cube.rs:22-23), rays that start ON a face, inside
    the sphere, exactly tangent: the decisions must still be the reference's"""
    rs = np.random.RandomState(5)
    n = 6000
    o = rs.uniform(-1.5, 1.5, (n, 3))
    d = rs.normal(size=(n, 3))
    ax = rs.randint(0, 3, n)
    d[np.arange(n), ax] = np.where(rs.rand(n) < 0.5, 0.0, -0.0)
    on_face = rs.rand(n) < 0.3
    o[on_face, ax[on_face]] = np.where(rs.rand(on_face.sum()) < 0.5, 0.5, -0.5)  # 0/0 for the cube on that axis
    tm, ti = np.full(n, 1e-12), np.full(n, INF)
    for name, shape, fn in (("cube", cube(), cube_intersect), ("sphere", sphere(), sphere_intersect)):
        hit, t, nrm = fn(o, d, tm, ti)
        for i in range(n):
            h0, t0, n0 = O.shape_intersect(shape, o[i], d[i])
            assert h0 == bool(hit[i]), (name, i)
            if h0:
                assert ulps(t[i], t0) <= 4 and np.abs(nrm[i] - n0).max() <= 1e-14, (name, i)


def test_triangle_and_its_tree_root_test_agree_with_an_independent_restatement():
    """Mesh of ONE triangle: KdTree::intersect's root test (kdtree.rs:129-135) over BoundingBox::intersect, then
    Triangle::intersect — barycentric accept decisions identical, time and the interpolated normal to a few ulp."""
    rs = np.random.RandomState(21)
    total_hits = 0
    for k in range(40):
        v = rs.uniform(-2.0, 2.0, (3, 3))
        if k % 5 == 0:
            v[:, k % 3] = rs.uniform(-1, 1)          # axis-aligned: a bounding box without thickness
        nn = unit(rs.normal(size=(3, 3)) * 0.3 + np.cross(v[1] - v[0], v[2] - v[0]))
        tri = Triangle(tuple(v[0]), tuple(v[1]), tuple(v[2]), tuple(nn[0]), tuple(nn[1]), tuple(nn[2]))
        mesh = Mesh([tri])
        n = 400
        o = rs.uniform(-4.0, 4.0, (n, 3))
        bary = rs.dirichlet((1.0, 1.0, 1.0), n) * rs.uniform(0.6, 1.6, (n, 1))  # targets inside and just outside
        target = bary @ v
        d = (target - o) * rs.uniform(0.3, 1.5, (n, 1))
        t_min = np.full(n, 1e-12)
        time = np.where(rs.rand(n) < 0.3, rs.uniform(0.2, 3.0, n), INF)
        hit, t, nrm, (u, vv, w) = triangle_intersect((v[0], v[1], v[2], nn[0], nn[1], nn[2]), o, d, t_min, time)
        box = np.concatenate([v.min(axis=0), v.max(axis=0)])
        b_min, b_max = bbox_intersect(box, o, d)
        enters = ~(np.fmax(b_min, t_min) > np.fmin(b_max, time))
        for i in range(n):
            bm0 = O.bbox_intersect(box, o[i], d[i])
            for mine_b, ref_b in ((b_min[i], bm0[0]), (b_max[i], bm0[1])):
                assert (np.isnan(mine_b) and np.isnan(ref_b)) or mine_b == ref_b, (k, i, mine_b, ref_b)  # divisions, min, max: exact
            h0, t0, n0 = O.shape_intersect(mesh, o[i], d[i], t_min[i], time[i])
            want = bool(hit[i] and enters[i])
            edge = min(abs(u[i]), abs(vv[i]), abs(w[i])) < 1e-12  # on an edge to rounding: either side is a valid reading
            assert h0 == want or edge, (k, i, h0, want, u[i], vv[i], w[i])
            if h0 and want:
                total_hits += 1
                assert ulps(t[i], t0) <= 8, (k, i, t[i], t0)
                assert np.abs(nrm[i] - n0).max() <= 1e-13, (k, i, nrm[i], n0)
    assert total_hits > 3000


def test_transformed_intersect_agrees_with_numpy_matrices():
    """Ray::apply_transform + Transformed::intersect (shape.rs:64-71, 128-137): the local ray through numpy's inverse,
    the normal through numpy's inverse transpose — rpt_amd/glm.py computes both by cofactors for the library."""
    rs = np.random.RandomState(31)
    worst = 0.0
    for k in range(60):
        shape = [sphere(), cube()][k % 2].scale(tuple(rs.uniform(0.4, 2.0, 3))).rotate(rs.uniform(0, 6.28), tuple(unit(rs.normal(size=3)))) \
            .translate(tuple(rs.uniform(-1.5, 1.5, 3)))
        M = np.array(shape.transform_m).reshape(4, 4).T          # column-major, like glm
        Minv = np.linalg.inv(M)
        NT = np.linalg.inv(M[:3, :3]).T
        n = 250
        o, d = random_rays(rs, n, reach=4.0)
        lo = (Minv @ np.concatenate([o, np.ones((n, 1))], axis=1).T).T[:, :3]
        ld = (Minv @ np.concatenate([d, np.zeros((n, 1))], axis=1).T).T[:, :3]
        fn = sphere_intersect if k % 2 == 0 else cube_intersect
        hit, t, nrm = fn(lo, ld, np.full(n, 1e-12), np.full(n, INF))
        wn = normalize((NT @ nrm.T).T)
        for i in range(n):
            h0, t0, n0 = O.shape_intersect(shape, o[i], d[i])
            graze = abs(t[i] - t0) > 1e-9 * max(1.0, abs(t0))  # a ray that grazes the shape may tip either way between two inverses
            assert h0 == bool(hit[i]) or graze or not np.isfinite(t[i]), (k, i)
            if h0 and hit[i] and not graze:
                assert abs(t[i] - t0) <= 1e-11 * max(1.0, abs(t0)) and np.abs(wn[i] - n0).max() <= 1e-9, (k, i, t[i], t0, wn[i], n0)
                worst = max(worst, abs(t[i] - t0) / max(1.0, abs(t0)))
    assert worst < 1e-11


# =============================================================================================== Shape::sample
def uniform_int(st, n):
    """rand 0.8.3 Uniform::from(0..n) for usize (UniformInt::sample, 64-bit): widening multiply with rejection zone"""
    rng = n
    zone = 0xFFFFFFFFFFFFFFFF - ((0xFFFFFFFFFFFFFFFF - rng + 1) % rng)
    while True:
        v = st.u64()
        m = v * rng
        hi, lo = m >> 64, m & 0xFFFFFFFFFFFFFFFF
        if lo <= zone:
            return hi


def sphere_sample(target, st):
    """Sphere::sample (sphere.rs:52-64)"""
    x, y = st.unit_disc()
    z = math.sqrt(1.0 - x * x - y * y)
    n = target / math.sqrt((target[0] * target[0] + target[1] * target[1]) + target[2] * target[2])
    is_normal = n[0] != 0.0 and abs(n[0]) >= 2.2250738585072014e-308 and math.isfinite(n[0])
    n1 = np.array([n[1], -n[0], 0.0]) if is_normal else np.array([0.0, -n[2], n[1]])
    n1 = n1 / math.sqrt((n1[0] * n1[0] + n1[1] * n1[1]) + n1[2] * n1[2])
    n2 = np.cross(n1, n)
    p = x * n1 + y * n2 + z * n
    return p, p, z * (1.0 / math.pi)


def cube_sample(target, st):
    """Cube::sample (cube.rs:74-87)"""
    a, b = st.f64() - 0.5, st.f64() - 0.5
    face = uniform_int(st, 6)
    v, n = [((a, b, 0.5), (0, 0, 1.0)), ((a, b, -0.5), (0, 0, -1.0)), ((a, 0.5, b), (0, 1.0, 0)), ((a, -0.5, b), (0, -1.0, 0)),
            ((0.5, a, b), (1.0, 0, 0)), ((-0.5, a, b), (-1.0, 0, 0))][face]
    return np.array(v), np.array(n, float), 1.0 / 6.0


def triangle_sample(tri, st):
    """Triangle::sample (mesh.rs:84-98)"""
    v1, v2, v3, n1, n2, n3 = tri
    u, v = st.f64(), st.f64()
    while u + v > 1.0:
        u, v = st.f64(), st.f64()
    w = 1.0 - u - v
    cr = np.cross(v2 - v1, v3 - v1)
    area = 0.5 * math.sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2])
    nn = u * n1 + v * n2 + w * n3
    return u * v1 + v * v2 + w * v3, nn / math.sqrt((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]), 1.0 / area


def transformed_sample(M, inner, target, st):
    """Transformed::sample (shape.rs:139-151) with numpy's matrices"""
    Minv, L = np.linalg.inv(M), M[:3, :3]
    NT = np.linalg.inv(L).T
    local_target = (Minv @ np.append(target, 1.0))[:3]
    v, n, p = inner(local_target, st)
    nn = NT @ n
    nn = nn / math.sqrt((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2])
    height = float(np.dot(L @ n, nn))
    base = np.linalg.det(L) / height
    return (M @ np.append(v, 1.0))[:3], nn, p / base


def close(a, b, tol):
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= tol * max(1.0, float(np.abs(np.asarray(b)).max()))


def test_shape_samples_agree_with_an_independent_restatement():
    """every Shape::sample of the closed set on the same Philox draws as the oracle: the same NUMBER of draws (the same
    rejections), the point, the normal and the density"""
    rs = np.random.RandomState(41)
    counts = {"sphere": 0, "cube": 0, "tri": 0, "group": 0, "xf": 0}
    for i in range(12000):
        seed, pixel, sample, draw0 = 500 + i, i % 1013, i // 7, int(rs.randint(0, 6))
        target = rs.uniform(-6, 6, 3)
        kind = i % 5
        st = Stream(seed, pixel, sample, draw0)
        if kind == 0:
            shape, mine = sphere(), sphere_sample(target, st)
            counts["sphere"] += 1
        elif kind == 1:
            shape, mine = cube(), cube_sample(target, st)
            counts["cube"] += 1
        elif kind == 2:  # a mesh of several triangles: KdTree::sample's uniform pick (kdtree.rs:138-143), then the triangle's
            m = int(rs.randint(1, 6))
            vs = rs.uniform(-2, 2, (m, 3, 3))
            ns = unit(rs.normal(size=(m, 3, 3)))
            tris = [Triangle(*(tuple(vs[j, c]) for c in range(3)), *(tuple(ns[j, c]) for c in range(3))) for j in range(m)]
            shape = Mesh(tris)
            j = uniform_int(st, m)
            v, n, p = triangle_sample((vs[j, 0], vs[j, 1], vs[j, 2], ns[j, 0], ns[j, 1], ns[j, 2]), st)
            mine = (v, n, p / m)
            counts["tri"] += 1
        elif kind == 3:  # a group of placed spheres and cubes: the pick, then Transformed::sample of the child
            m = int(rs.randint(1, 5))
            kids, fns = [], []
            for j in range(m):
                base, fn = (sphere(), sphere_sample) if rs.rand() < 0.5 else (cube(), cube_sample)
                kid = base.scale(tuple(rs.uniform(0.3, 1.5, 3))).rotate_y(rs.uniform(0, 3)).translate(tuple(rs.uniform(-2, 2, 3)))
                kids.append(kid)
                fns.append((np.array(kid.transform_m).reshape(4, 4).T, fn))
            shape = KdTree(kids)
            j = uniform_int(st, m)
            v, n, p = transformed_sample(fns[j][0], fns[j][1], target, st)
            mine = (v, n, p / m)
            counts["group"] += 1
        else:
            base, fn = (sphere(), sphere_sample) if i % 2 else (cube(), cube_sample)
            shape = base.scale(tuple(rs.uniform(0.3, 2.0, 3))).rotate(rs.uniform(0, 6), tuple(unit(rs.normal(size=3)))).translate(tuple(rs.uniform(-3, 3, 3)))
            mine = transformed_sample(np.array(shape.transform_m).reshape(4, 4).T, fn, target, st)
            counts["xf"] += 1
        v0, n0, p0, draw_o = O.shape_sample(shape, target, seed=seed, pixel=pixel, sample=sample, draw=draw0)
        assert st.draw == draw_o, (i, kind, st.draw, draw_o)
        tol = 1e-11 if kind >= 3 else 4e-15   # (two different matrix inverses behind the transformed ones)
        assert close(mine[0], v0, tol) and close(mine[1], n0, tol) and abs(mine[2] - p0) <= 10 * tol * abs(p0), (i, kind, mine, (v0, n0, p0))
    assert min(counts.values()) >= 2000


# =============================================================================================== camera, environment, clamp
def gen_range(st, lo, hi):
    """rand 0.8.3 Rng::gen_range(lo..hi) for f64 — UniformFloat::sample_single: value1_2 from the 52 high bits, then
    (value1_2 - 1.0) * (hi - lo) + lo, redrawn while the result is not below hi"""
    scale = hi - lo
    while True:
        bits = (st.u64() >> 12) | 0x3FF0000000000000
        v12 = np.frombuffer(np.uint64(bits).tobytes(), dtype=np.float64)[0]
        res = (v12 - 1.0) * scale + lo
        if res < hi:
            return res


def camera_ray(cam, width, height, x, y, st):
    """Renderer::get_color's pixel mapping and jitter (renderer.rs:131-139), then Camera::cast_ray (camera.rs:64-81)"""
    dim = float(max(width, height))
    xn = (float(2 * x + 1) - float(width)) / dim
    yn = (float(2 * (height - y) - 1) - float(height)) / dim
    dx = gen_range(st, -1.0 / dim, 1.0 / dim)
    dy = gen_range(st, -1.0 / dim, 1.0 / dim)
    px, py = xn + dx, yn + dy
    eye, direction, up = (np.array(v, float) for v in (cam.eye, cam.direction, cam.up))
    dd = 1.0 / math.tan(cam.fov / 2.0)
    right = np.cross(direction, up)
    right = right / math.sqrt((right[0] * right[0] + right[1] * right[1]) + right[2] * right[2])
    origin = eye.copy()
    new_dir = dd * direction + px * right + py * up
    if cam.aperture > 0.0:
        nd = new_dir / math.sqrt((new_dir[0] * new_dir[0] + new_dir[1] * new_dir[1]) + new_dir[2] * new_dir[2])
        focal_point = origin + nd * cam.focal_distance
        lx, ly = st.unit_disc()
        origin = origin + (lx * right + ly * up) * cam.aperture
        new_dir = focal_point - origin
    return origin, new_dir / math.sqrt((new_dir[0] * new_dir[0] + new_dir[1] * new_dir[1]) + new_dir[2] * new_dir[2])


def test_camera_rays_agree_with_an_independent_restatement():
    rs = np.random.RandomState(51)
    for k in range(24):
        eye = rs.uniform(-5, 5, 3)
        cam = Camera.look_at(tuple(eye), tuple(rs.uniform(-1, 1, 3)), (0.0, 1.0, 0.0), float(rs.uniform(0.3, 1.4)))
        if k % 2:
            cam = cam.focus(tuple(rs.uniform(-1, 1, 3)), float(rs.uniform(0.01, 0.3)))
        w, h = int(rs.randint(3, 200)), int(rs.randint(3, 200))
        p = make_params(w, h, 2, 1, seed=900 + k)
        for _ in range(500):
            x, y, s = int(rs.randint(0, w)), int(rs.randint(0, h)), int(rs.randint(0, 50))
            o0, d0 = O.camera_ray(cam, p, x, y, s)
            o1, d1 = camera_ray(cam, w, h, x, y, Stream(900 + k, y * w + x, s, 0))
            assert np.abs(o1 - o0).max() <= 4e-15 * max(1.0, np.abs(o0).max()), (k, x, y, s, o1, o0)
            assert np.abs(d1 - d0).max() <= 4e-15, (k, x, y, s, d1, d0)


def hdri_color(tex, dirv):
    """Hdri::get_color + bilinear_sample (environment.rs:25-52); `as u32` saturates, glm::mix(a, b, t) = a (1 - t) + b t"""
    h, w = tex.shape[:2]
    dn = dirv / math.sqrt((dirv[0] * dirv[0] + dirv[1] * dirv[1]) + dirv[2] * dirv[2])
    azimuth = math.atan2(dn[2], dn[0]) + math.pi
    polar = math.acos(dn[1])
    x = azimuth / (2.0 * math.pi) * float(w - 1)
    y = polar / math.pi * float(h - 1)
    x0, y0 = min(int(x), w - 1), min(int(y), h - 1)
    ax, ay = x - float(x0), y - float(y0)
    flat = tex.reshape(-1, 3)

    def texel(j):
        return flat[j]

    def mix(a, b, t):
        return a * (1.0 - t) + b * t
    top = mix(texel(y0 * w + x0), texel(y0 * w + x0 + 1), ax)
    bot = mix(texel((y0 + 1) * w + x0), texel((y0 + 1) * w + x0 + 1), ax)
    return mix(top, bot, ay)


def test_hdri_lookup_agrees_with_an_independent_restatement():
    rs = np.random.RandomState(61)
    tex = scenes.synthetic_hdri(48, 24, seed=3)
    env = Environment.Hdri(tex)
    arr = np.asarray(tex.buf, float).reshape(tex.height, tex.width, 3)
    n_ok = n_oob = 0
    for i in range(10000):
        d = rs.normal(size=3) * rs.uniform(0.2, 3.0)
        if i % 50 == 0:
            d[rs.randint(0, 3)] = 0.0
        # the last row / column reads one texel past (environment.rs:45-48 indexes y0 + 1, x0 + 1): directions that land
        # exactly there panic in the reference; the random ones here do not reach it
        c0 = O.env_color(env, d)
        try:
            c1 = hdri_color(arr, d)
        except IndexError:  # straight down: polar = pi lands ON the last row and row y0 + 1 does not exist — the reference
            n_oob += 1      # panics there (the library and the oracle read zeros, oracle.cpp Env::px); nothing to compare
            continue
        # atan2 / acos are glibc's here and include/rpt_math.h's there: a last-bit difference in the angle moves the
        # bilinear weights by ~1e-16 * width
        assert np.abs(c1 - c0).max() <= 1e-12 * max(1.0, np.abs(c0).max()), (i, d, c1, c0)
        n_ok += 1
    assert n_ok + n_oob == 10000 and n_oob <= 20, (n_ok, n_oob)


def test_firefly_clamp_fold_agrees_with_an_independent_restatement():
    """trace_ray's unwinding (renderer.rs:152-167): L_k = A_k + min(1/pdf_k * (f_k . L_{k+1}) * |wi_k . n_k|, 100) per
    channel, from the path's last vertex back to the camera — recomputed here from the oracle's own per-depth terms and
    compared with the radiance it returns for the sample.  The terms come from a scene with fireflies: a bright lamp that
    BSDF-sampled bounces run into (its emitted radiance is far above the clamp), behind glass and next to a rough mirror."""
    from rpt_amd import Light, Material, Object, Scene, hex_color
    scene = Scene()
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Object(sphere().translate((0.0, 0.0, 0.0))).material(Material.clear(1.5, 0.05)))
    scene.add(Object(cube().scale((0.5, 0.5, 0.5)).translate((1.6, -0.5, 0.3))).material(Material.metallic_(hex_color(0xFFD080), 0.1)))
    lamp = Object(sphere().scale((0.9, 0.9, 0.9)).translate((0.5, 3.0, 1.0))).material(Material.light((1.0, 0.9, 0.8), 600.0))
    scene.add(lamp)                 # visible: a bounce that runs into it picks up 600 (examples/cornell.rs adds both, too)
    scene.add(Light.Object(lamp))
    cam = Camera.look_at((0.0, 1.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.6)
    p = make_params(48, 36, 6, 1, seed=77)
    osc = O.OracleScene(scene)
    clamped = paths = 0
    for k in range(4000):
        x, y, s = k % 48, (k // 48) % 36, k // (48 * 36)
        L, rec = osc.trace_sample(cam, p, x, y, s)   # rec[k] = (A_k[3], f_k[3], 1/pdf_k, |wi_k . n_k|) per depth; the last one has A only
        acc = rec[-1][:3].copy()
        for r in rec[-2::-1]:
            indirect = r[6] * (r[3:6] * acc) * r[7]
            clamped += int((indirect > 100.0).any())
            acc = r[:3] + np.minimum(indirect, 100.0)
        paths += 1
        assert (acc == L).all(), (k, acc, L)   # the same operations in the same order: the same bits
    assert paths == 4000 and clamped > 20
