"""Known-answer tests that pin the CPU oracle to the reference's formulas.

The reference has no tests on this path (only `colors_work`), so every expected
value below is derived by hand from the cited reference lines.  These run on CPU only.
"""
import math

import numpy as np
import pytest

import rpt_amd
from rpt_amd import Camera, Light, Material, Object, Triangle, cube, make_params, plane, polygon, sphere
from rpt_amd import hex_color

INF = float("inf")


# ---------------------------------------------------------------- RNG (Philox + rand/rand_distr)
def test_philox4x32_10_random123_known_answers(oracle):
    # Random123 kat_vectors for philox4x32-10
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_stream_layout(oracle):
    # draw 2b and 2b+1 are the two halves of block b with counter (pixel, sample lo, sample hi, b)
    seed, pixel, sample = 0x0123456789ABCDEF, 77, (5 << 32) | 9
    for b in range(3):
        w = oracle.philox([pixel, 9, 5, b], [seed & 0xFFFFFFFF, seed >> 32])
        assert oracle.rng_u64(seed, pixel, sample, 2 * b) == (w[1] << 32) | w[0]
        assert oracle.rng_u64(seed, pixel, sample, 2 * b + 1) == (w[3] << 32) | w[2]


def test_distributions_follow_rand_0_8(oracle):
    seed = 99
    for draw in range(0, 40, 2):
        u = oracle.rng_u64(seed, 1, 2, draw)
        v, d = oracle.rng_sample(0, seed=seed, pixel=1, sample=2, draw=draw)  # gen::<f64>()
        assert v[0] == (u >> 11) * 2.0 ** -53 and d == draw + 1
        v, d = oracle.rng_sample(1, -0.25, 0.75, seed=seed, pixel=1, sample=2, draw=draw)  # gen_range
        assert v[0] == ((u >> 12) * 2.0 ** -52) * 1.0 + -0.25
        v, d = oracle.rng_sample(2, 0.3, seed=seed, pixel=1, sample=2, draw=draw)  # gen_bool(0.3)
        assert bool(v[0]) == (u < int(0.3 * 2.0 ** 64))
        v, d = oracle.rng_sample(3, 6, seed=seed, pixel=1, sample=2, draw=draw)  # Uniform 0..6
        assert v[0] == (u * 6) >> 64 or d > draw + 1
    v, d = oracle.rng_sample(2, 1.0, seed=seed, draw=5)  # gen_bool(1.0): no draw consumed
    assert v[0] == 1.0 and d == 5


def test_unit_disc_and_circle(oracle):
    pts, d = [], 0
    for _ in range(4000):
        v, d = oracle.rng_sample(4, seed=5, draw=d)
        pts.append(v.copy())
    pts = np.array(pts)
    assert ((pts ** 2).sum(axis=1) <= 1.0).all()
    assert abs((pts ** 2).sum(axis=1).mean() - 0.5) < 0.02  # uniform disc: E[r^2] = 1/2
    assert d > 2 * 4000  # rejection consumed extra draws
    cs, d = [], 0
    for _ in range(2000):
        v, d = oracle.rng_sample(5, seed=6, draw=d)
        cs.append(v.copy())
    cs = np.array(cs)
    assert np.allclose((cs ** 2).sum(axis=1), 1.0, atol=1e-12)
    assert abs(cs.mean(axis=0)).max() < 0.06


# ---------------------------------------------------------------- shapes
def test_sphere_known_answers(oracle):
    hit, t, n = oracle.shape_intersect(sphere(), (0, 0, 10), (0, 0, -1))  # sphere.rs:13-45
    assert hit and t == 9.0 and tuple(n) == (0.0, 0.0, 1.0)
    hit, t, n = oracle.shape_intersect(sphere(), (0, 0, 0), (0, 0, 1))  # inside: far root, OUTWARD normal
    assert hit and t == 1.0 and tuple(n) == (0.0, 0.0, 1.0)
    hit, t, n = oracle.shape_intersect(sphere(), (0, 2, 10), (0, 0, -1))  # miss
    assert not hit and t == INF
    hit, t, n = oracle.shape_intersect(sphere(), (0, 0, 10), (0, 0, -1), t_min=9.5)  # t_min picks far root
    assert hit and t == 11.0 and tuple(n) == (0.0, 0.0, -1.0)
    hit, t, n = oracle.shape_intersect(sphere(), (0, 0, 10), (0, 0, -1), time=5.0)  # behind current best
    assert not hit and t == 5.0
    hit, t, n = oracle.shape_intersect(sphere(), (0, 0, 10), (0, 0, -2))  # non-unit dir: t halves
    assert hit and t == 4.5


def test_plane_known_answers(oracle):
    p = plane((0, 1, 0), -1.0)  # plane.rs:17-32
    hit, t, n = oracle.shape_intersect(p, (0, 1, 0), (0, -1, 0))
    assert hit and t == 2.0 and tuple(n) == (0.0, 1.0, 0.0)  # normal faces the ray
    hit, t, n = oracle.shape_intersect(p, (0, -3, 0), (0, 1, 0))
    assert hit and t == 2.0 and tuple(n) == (0.0, -1.0, 0.0)
    hit, _, _ = oracle.shape_intersect(p, (0, 1, 0), (1, 0, 0))  # parallel: |cos| < 1e-8
    assert not hit
    hit, _, _ = oracle.shape_intersect(p, (0, 1, 0), (1, -0.9e-8, 0))
    assert not hit
    hit, _, _ = oracle.shape_intersect(p, (0, 1, 0), (0, 1, 0))  # behind
    assert not hit
    hit, t, n = oracle.shape_intersect(plane((0, 2, 0), -2.0), (0, 1, 0), (0, -1, 0))  # unnormalised normal
    assert hit and t == 2.0 and tuple(n) == (0.0, 1.0, 0.0)


def test_cube_known_answers(oracle):
    c = cube()  # cube.rs:20-72
    hit, t, n = oracle.shape_intersect(c, (0, 0, 5), (0, 0, -1))
    assert hit and t == 4.5 and tuple(n) == (0.0, 0.0, 1.0)
    hit, t, n = oracle.shape_intersect(c, (0, 0, 0), (1, 0, 0))  # from inside: exit face
    assert hit and t == 0.5 and tuple(n) == (1.0, 0.0, 0.0)
    hit, t, n = oracle.shape_intersect(c, (5, 0.25, 0.1), (-1, 0, 0))
    assert hit and t == 4.5 and tuple(n) == (1.0, 0.0, 0.0)
    hit, _, _ = oracle.shape_intersect(c, (5, 0.75, 0), (-1, 0, 0))
    assert not hit
    hit, _, _ = oracle.shape_intersect(c, (0, 0, 5), (0, 0, 1))  # box behind
    assert not hit


def test_triangle_known_answers(oracle):
    tri = polygon([(0, 0, 0), (1, 0, 0), (0, 1, 0)])  # one triangle, normal +z (mesh.rs:26-36)
    hit, t, n = oracle.shape_intersect(tri, (0.25, 0.25, 1), (0, 0, -1))
    assert hit and t == 1.0 and tuple(n) == (0.0, 0.0, 1.0)  # normal NOT flipped towards the ray
    hit, t, n = oracle.shape_intersect(tri, (0.25, 0.25, -1), (0, 0, 1))
    assert hit and t == 1.0 and tuple(n) == (0.0, 0.0, 1.0)
    hit, _, _ = oracle.shape_intersect(tri, (0.5, 0.5, 1), (0, 0, -1))  # on the hypotenuse: u == 0 is inside
    assert hit
    hit, _, _ = oracle.shape_intersect(tri, (0.5, 0.5 + 1e-12, 1), (0, 0, -1))
    assert not hit
    # quirk (kdtree.rs:54-68 with f64::min/max ignoring NaN): a ray lying exactly in a face plane
    # of the mesh's bounding box with a zero direction component there gives (0/0 = NaN, inf) ->
    # interval [inf, inf] -> the KdTree root test rejects it, although the triangle edge is inclusive
    hit, _, _ = oracle.shape_intersect(tri, (0.5, 0.0, 1), (0, 0, -1))
    assert not hit
    hit, _, _ = oracle.shape_intersect(tri, (0.5, 0.0, 1), (0, 1e-300, -1))  # any non-zero dy: edge hit
    assert hit
    hit, _, _ = oracle.shape_intersect(tri, (0.75, 0.75, 1), (0, 0, -1))  # outside
    assert not hit
    hit, _, _ = oracle.shape_intersect(tri, (0.25, 0.25, 1), (0, 0, -1), time=1.0)  # t >= record.time rejected
    assert not hit
    hit, _, _ = oracle.shape_intersect(tri, (0.25, 0.25, 1), (1, 0, 0))  # parallel
    assert not hit


def test_smooth_normal_interpolation(oracle):
    t = Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 0), (0, 1, 0))
    mesh = rpt_amd.Mesh([t])
    hit, tt, n = oracle.shape_intersect(mesh, (0.25, 0.5, 1), (0, 0, -1))
    # mesh.rs:64-77: v=0.25 (towards v2), w=0.5 (towards v3), u=0.25 -> normalize(u*n1+v*n2+w*n3)
    e = np.array([0.25, 0.5, 0.25]) / math.sqrt(0.25 ** 2 + 0.5 ** 2 + 0.25 ** 2)
    assert hit and tt == 1.0 and np.allclose(n, e, atol=1e-15)


def test_transformed_preserves_t_and_maps_normals(oracle):
    # shape.rs:128-137: t is shared across spaces (dir not renormalised); normal by inverse-transpose
    s = sphere().scale((2, 2, 2)).translate((0, 0, -5))
    hit, t, n = oracle.shape_intersect(s, (0, 0, 5), (0, 0, -1))
    assert hit and t == 8.0 and tuple(n) == (0.0, 0.0, 1.0)
    e = sphere().scale((1, 2, 1))  # ellipsoid: normal at (x,y) ~ (x, y/4)
    x, y = 0.6, 1.6  # on the ellipse x^2 + (y/2)^2 = 1
    hit, t, n = oracle.shape_intersect(e, (x, y, 5), (0, 0, -1))
    en = np.array([x, y / 4.0, 0.0])
    assert hit and abs(t - 5.0) < 1e-7 and np.allclose(n[:2] / np.linalg.norm(n[:2]), (en / np.linalg.norm(en))[:2], atol=1e-6)
    r = cube().rotate_y(math.pi / 4).translate((0, 0, 0))
    hit, t, n = oracle.shape_intersect(r, (0, 0, 5), (0, 0, -1))
    assert hit and abs(t - (5 - math.sqrt(0.5))) < 1e-12
    assert abs(abs(n[0]) - math.sqrt(0.5)) < 1e-12 and abs(n[2] - math.sqrt(0.5)) < 1e-12


def test_bbox_slab_with_zero_direction_components(oracle):
    box = (-1, -1, -1, 1, 1, 1)  # kdtree.rs:54-68: true divisions, NaN-ignoring min/max
    assert oracle.bbox_intersect(box, (0, 0, 5), (0, 0, -1)) == (4.0, 6.0)
    assert oracle.bbox_intersect(box, (0, 0, 0), (1, 0, 0)) == (-1.0, 1.0)
    a, b = oracle.bbox_intersect(box, (2, 0, 5), (0, 0, -1))  # outside in x with dx = 0: empty interval
    assert a > b
    a, b = oracle.bbox_intersect(box, (1, 0, 5), (0, 0, -1))  # on the face with dx = 0: (-inf, NaN) ->
    assert b == -INF and a > b                                  # min = max = -inf: empty (reference quirk)


def test_shape_sample(oracle):
    # Sphere::sample sphere.rs:52-64: cosine-weighted hemisphere facing the target, pdf z/pi
    target = (0.0, 0.0, 7.0)
    d = 0
    zs = []
    for _ in range(3000):
        v, n, p, d = oracle.shape_sample(sphere(), target, seed=3, draw=d)
        assert abs(np.dot(v, v) - 1.0) < 1e-12 and (v == n).all()
        assert v[2] >= 0.0 and abs(p - v[2] / math.pi) < 1e-15
        zs.append(v[2])
    assert abs(np.mean(zs) - 2.0 / 3.0) < 0.02  # E[cos] under cosine weighting
    # Triangle / KdTree::sample: uniform object pick, pdf = 1/area/num  (mesh.rs:84-98, kdtree.rs:138-143)
    quad = polygon([(0, 0, 0), (2, 0, 0), (2, 3, 0), (0, 3, 0)])
    v, n, p, d2 = oracle.shape_sample(quad, (0, 0, 1), seed=4)
    assert abs(p - (1.0 / 3.0) / 2.0) < 1e-15 and v[2] == 0.0 and tuple(n) == (0.0, 0.0, 1.0)
    # Transformed::sample shape.rs:139-151: uniform scale s divides the pdf by s^2
    v, n, p, _ = oracle.shape_sample(sphere().scale((2, 2, 2)).translate((0, 12, 0)), (0, 0, 0), seed=9)
    v0, n0, p0, _ = oracle.shape_sample(sphere(), (0, -6, 0), seed=9)
    assert np.allclose(v, np.array(v0) * 2 + (0, 12, 0)) and abs(p - p0 / 4.0) < 1e-15
    # Cube::sample cube.rs:74-87
    v, n, p, _ = oracle.shape_sample(cube(), (0, 0, 0), seed=11)
    assert p == 1.0 / 6.0 and np.abs(v).max() == 0.5 and abs(np.dot(v, n) - 0.5) < 1e-15
    with pytest.raises(rpt_amd.RptGpuError):  # Plane::sample is unimplemented!() plane.rs:34-36
        oracle.shape_sample(plane((0, 1, 0), 0), (0, 0, 0))


# ---------------------------------------------------------------- material
def test_bsdf_known_answers(oracle):
    n = (0.0, 0.0, 1.0)
    up = (0.0, 0.0, 1.0)
    diffuse = Material.diffuse((0.5, 0.25, 0.125))
    # opaque, below the horizon -> 0 (material.rs:130-133); +0.0 counts as outside
    assert tuple(oracle.bsdf(diffuse, n, up, (0, 0, -1))) == (0.0, 0.0, 0.0)
    assert tuple(oracle.bsdf(diffuse, n, (0, 0.6, -0.8), up)) == (0.0, 0.0, 0.0)
    # normal incidence, wi = wo = n: h = n, nh = 1, D = 1/(pi m^2), F = F0 = 0.04, G = 1
    f = oracle.bsdf(diffuse, n, up, up)
    F0 = ((1.5 - 1) / (1.5 + 1)) ** 2
    spec = (1.0 / math.pi) * F0 / 4.0
    exp = spec + (1 - F0) * np.array(diffuse.color) / math.pi
    assert np.allclose(f, exp, rtol=1e-14)
    # metallic: F0 = color, no change to the Lambert term formula (1-F)*c/pi
    metal = Material.metallic_((0.9, 0.5, 0.1), 0.5)
    f = oracle.bsdf(metal, n, up, up)
    c = np.array(metal.color)
    assert np.allclose(f, (1 / (math.pi * 0.25)) * c / 4.0 + (1 - c) * c / math.pi, rtol=1e-14)
    # transparent, same side: specular only (material.rs:166-167)
    glass = Material.clear(1.5, 0.5)
    f = oracle.bsdf(glass, n, up, up)
    assert np.allclose(f, (1 / (math.pi * 0.25)) * F0 / 4.0, rtol=1e-14)
    # transparent, opposite sides, straight through: h = normalize(wi*eta + wo) with wi = -n
    f = oracle.bsdf(glass, n, up, (0, 0, -1))
    eta = 1.5
    h = -1.0  # (wi*eta + wo) = (0,0,-0.5) -> h = (0,0,-1)
    wih, woh, nh = 1.0, -1.0, -1.0
    D = 1 / (0.25 * math.pi)
    F = F0  # (1-|wi.h|)^5 = 0
    G = min(1.0, 2 * 1.0 / 1.0)
    exp = abs(wih * woh / (-1.0 * 1.0)) * (D * (1 - F) * G / (eta * wih + woh) ** 2)
    assert np.allclose(f, exp, rtol=1e-14)
    # total internal reflection branch: F = 1 when both inside and sin*index > 1 (material.rs:147-149)
    wo_in = np.array([0.8, 0.0, -0.6])
    wi_in = np.array([-0.8, 0.0, -0.6])
    f = oracle.bsdf(glass, n, wo_in, wi_in)
    nh2 = 1.0
    G = min(1.0, 2 * min(0.6, 0.6) / 0.6)
    assert np.allclose(f, D * 1.0 * G / (4 * 0.36), rtol=1e-13)


def test_sample_f_is_consistent_with_bsdf(oracle):
    # E[f * |cos| / pdf] over sample_f == integral of f*cos over the hemisphere (white-furnace
    # style check of material.rs:224-313 against material.rs:125-210)
    n = (0.0, 0.0, 1.0)
    wo = np.array([0.6, 0.0, 0.8])
    for mat in (Material.diffuse((0.8, 0.8, 0.8)), Material.specular((0.7, 0.3, 0.2), 0.5), Material()):
        est, d = np.zeros(3), 0
        N = 20000
        for _ in range(N):
            some, wi, pdf, d = oracle.sample_f(mat, n, wo, seed=21, draw=d)
            assert some and pdf > 0
            est += oracle.bsdf(mat, n, wo, wi) * abs(wi[2]) / pdf
        est /= N
        # quadrature over the upper hemisphere
        nt, nphi = 200, 400
        ct = (np.arange(nt) + 0.5) / nt
        ph = (np.arange(nphi) + 0.5) / nphi * 2 * math.pi
        quad = np.zeros(3)
        for c in ct:
            s = math.sqrt(1 - c * c)
            for p in ph[::4]:
                quad += oracle.bsdf(mat, n, wo, (s * math.cos(p), s * math.sin(p), c)) * c
        quad *= (1.0 / nt) * (2 * math.pi / (nphi // 4))
        assert np.allclose(est, quad, rtol=0.04), (est, quad)


def test_sample_f_draw_order_and_tir(oracle):
    n, wo = (0.0, 0.0, 1.0), (0.0, 0.0, 1.0)
    mat = Material.diffuse((0.5, 0.5, 0.5))
    # gen_bool(f) first with f = 0.8*0.04+0.2 = 0.232 (material.rs:233-235,264)
    u = oracle.rng_u64(5, 0, 0, 0)
    some, wi, pdf, d = oracle.sample_f(mat, n, wo, seed=5)
    took_specular = u < int(0.232 * 2.0 ** 64)
    if not took_specular:  # Malley: wi.z = sqrt(1-x^2-y^2), pdf includes both lobes
        assert wi[2] >= 0 and pdf >= (1 - 0.232) * wi[2] / math.pi
    # TIR in the transmitted branch returns None (material.rs:281-285): from inside at a grazing angle
    glass = Material.clear(1.5, 0.0001)
    wo_in = np.array([0.95, 0.0, -math.sqrt(1 - 0.95 ** 2)])
    nones = 0
    for s in range(200):
        some, wi, pdf, d = oracle.sample_f(glass, n, wo_in, seed=s)
        nones += (not some)
    assert nones > 100  # most draws pick the transmitted lobe (1-f = 0.768) and all of those are TIR


# ---------------------------------------------------------------- lights / camera / environment
def test_illuminate_point_and_directional(oracle):
    i, wi, dist, d = oracle.illuminate(Light.Point((100, 50, 25), (0, 5, 0)), (0, 0, 0))
    assert tuple(i) == (4.0, 2.0, 1.0) and tuple(wi) == (0.0, 1.0, 0.0) and dist == 5.0 and d == 0
    i, wi, dist, d = oracle.illuminate(Light.Directional((0.6, 0.6, 0.6), (0, -2, 0)), (1, 2, 3))
    assert tuple(i) == (0.6, 0.6, 0.6) and tuple(wi) == (0.0, 1.0, 0.0) and dist == INF and d == 0


def test_illuminate_far_sphere_light_expectation(oracle):
    # light.rs:34-45 + sphere.rs:52-64: E[intensity] -> emittance*color*pi*r^2/dist^2 for a far light
    r, dist, e = 2.0, 200.0, 40.0
    light = Light.Object(Object(sphere().scale((r, r, r)).translate((0, dist, 0))).material(Material.light((1, 1, 1), e)))
    acc, d = 0.0, 0
    N = 4000
    for _ in range(N):
        i, wi, dl, d = oracle.illuminate(light, (0, 0, 0), seed=8, draw=d)
        acc += i[0]
        assert abs(dl - dist) < r + 1e-9
    assert abs(acc / N - e * math.pi * r * r / dist ** 2) / (e * math.pi * r * r / dist ** 2) < 0.02


def test_camera_ray_and_pixel_mapping(oracle):
    # renderer.rs:132-134: dim = max(W,H); the fov spans the longer side; y = 0 is the top row
    cam = Camera()  # eye (0,0,10), dir -z, up +y, fov pi/6
    p = make_params(4, 2, 0, 1, seed=1)
    o, d = oracle.camera_ray(cam, p, 0, 0, 0)
    assert tuple(o) == (0.0, 0.0, 10.0) and abs(np.linalg.norm(d) - 1) < 1e-15
    cot = 1.0 / math.tan(math.pi / 12)
    # pixel (0,0): xn = (1-4)/4 = -0.75, yn = (2*2-1-2)/4 = 0.25, jitter within +-1/dim
    x_over_z = d[0] / -d[2] * cot
    y_over_z = d[1] / -d[2] * cot
    assert -0.75 - 0.25 <= x_over_z < -0.75 + 0.25 and 0.25 - 0.25 <= y_over_z < 0.25 + 0.25
    o, d = oracle.camera_ray(cam, p, 3, 1, 0)
    assert d[0] > 0 and d[1] < 0.05
    # look_at orthonormalises up (camera.rs:43-54)
    c2 = Camera.look_at((0, 0, 5), (0, 1, 0), (0, 1, 0), 0.5)
    assert abs(np.dot(c2.up, c2.direction)) < 1e-15 and abs(np.linalg.norm(c2.up) - 1) < 1e-15
    c3 = Camera.look_at((0, 0, 5), (0, 0, 0), (0, 1, 0), 0.5).focus((0, 0, 1), 0.1)
    assert c3.focal_distance == 4.0 and c3.aperture == 0.1


def test_environment_lookup(oracle):
    env = rpt_amd.Environment.Color((0.1, 0.2, 0.3))
    assert tuple(oracle.env_color(env, (0, 1, 0))) == (0.1, 0.2, 0.3)
    # 4x3 HDRI with texel value = column index + 10*row (environment.rs:25-52)
    w, h = 4, 3
    buf = np.zeros((h, w, 3))
    for yy in range(h):
        for xx in range(w):
            buf[yy, xx] = xx + 10 * yy
    env = rpt_amd.Environment.Hdri(rpt_amd.Hdri(w, h, buf))
    # dir = -x: azimuth = atan2(0,-1)+pi = 2pi -> x = 3 (clamped x0 = 3, ax = 0), polar = pi/2 -> y = 1
    c = oracle.env_color(env, (-1, 0, 0))
    assert np.allclose(c, 3 + 10, atol=1e-12)
    # dir = +x: azimuth = pi -> x = 1.5, polar pi/2 -> y = 1 : mix of columns 1,2 on row 1
    c = oracle.env_color(env, (1, 0, 0))
    assert np.allclose(c, 1.5 + 10, atol=1e-12)
    # straight up: polar 0 -> row 0
    c = oracle.env_color(env, (1e-9, 1, 0))
    assert np.allclose(c, 1.5, atol=1e-6)


# ---------------------------------------------------------------- the estimator
def _furnace_scene(color=(1.0, 1.0, 1.0)):
    scene = rpt_amd.Scene()
    scene.environment = rpt_amd.Environment.Color(color)
    return scene


def test_miss_returns_environment_unclamped(oracle):
    scene = _furnace_scene((500.0, 2.0, 3.0))  # camera-ray miss is NOT clamped (renderer.rs:147)
    img = oracle.OracleScene(scene).render(Camera(), make_params(4, 3, 5, 3), threads=1)
    assert (img == np.array([500.0, 2.0, 3.0])).all()
    img = oracle.OracleScene(scene).render(Camera(), make_params(4, 3, 5, 3, exposure_value=1.0), threads=1)
    assert (img == np.array([1000.0, 4.0, 6.0])).all()  # * 2^EV (renderer.rs:141)


def test_nested_firefly_clamp_and_records(oracle):
    # a bright environment behind a diffuse plane: every indirect term is clamped to 100 per
    # channel at every depth (renderer.rs:162-167); L_0 = A_0 + min(W_0 * L_1, 100)
    scene = _furnace_scene((1e6, 1e6, 1e6))
    scene.add(Object(plane((0, 0, 1), 0.0)).material(Material.diffuse((0.5, 0.5, 0.5))))
    osc = oracle.OracleScene(scene)
    p = make_params(2, 2, 3, 1, seed=2)
    rgb, rec = osc.trace_sample(Camera(), p, 0, 0, 0)
    assert len(rec) >= 2
    L = rec[-1][:3].copy()
    for k in range(len(rec) - 2, -1, -1):
        A, f, inv_pdf, abscos = rec[k][:3], rec[k][3:6], rec[k][6], rec[k][7]
        L = A + np.minimum(inv_pdf * (f * L) * abscos, 100.0)
    assert (L == rgb).all()
    assert (rgb <= 100.0 + 1e-9).all() and rgb.max() == 100.0  # no lights: A_0 = 0, indirect clamped


def test_emission_ambient_and_invisible_lights(oracle):
    # emission is added at every hit (renderer.rs:153); ambient adds ambient*albedo with no
    # visibility (renderer.rs:187-188); light geometry is not in scene.objects (scene.rs:9-12)
    scene = rpt_amd.Scene()
    scene.add(Object(plane((0, 0, 1), 0.0)).material(Material(color=(0.5, 0.25, 1.0), emittance=2.0, roughness=1.0)))
    scene.add(Light.Ambient((0.1, 0.2, 0.3)))
    img = oracle.OracleScene(scene).render(Camera(), make_params(2, 2, 0, 2), threads=1)
    exp = np.array([2.0 * 0.5 + 0.1 * 0.5, 2.0 * 0.25 + 0.2 * 0.25, 2.0 * 1.0 + 0.3 * 1.0])
    assert np.allclose(img, exp, rtol=1e-15)
    scene2 = rpt_amd.Scene()
    scene2.add(Light.Object(Object(sphere()).material(Material.light((1, 1, 1), 10.0))))
    img = oracle.OracleScene(scene2).render(Camera(), make_params(4, 4, 2, 2), threads=1)
    assert (img == 0.0).all()  # the light sphere in front of the camera is invisible


def test_point_light_direct_illumination_closed_form(oracle):
    # diffuse plane z=0, point light at (0,0,2), camera straight down at the origin, B=0:
    # L = bsdf(n, wo, wi) * color/len^2 * (wi.n) with wi = wo = n  (renderer.rs:198-199)
    scene = rpt_amd.Scene()
    alb = (0.5, 0.5, 0.5)
    scene.add(Object(plane((0, 0, 1), 0.0)).material(Material.diffuse(alb)))
    scene.add(Light.Point((8.0, 8.0, 8.0), (0, 0, 2)))
    cam = Camera(eye=(0, 0, 10), direction=(0, 0, -1), up=(0, 1, 0), fov=1e-6)
    img = oracle.OracleScene(scene).render(cam, make_params(1, 1, 0, 1), threads=1)
    f = oracle.bsdf(Material.diffuse(alb), (0, 0, 1), (0, 0, 1), (0, 0, 1))
    assert np.allclose(img[0], f * 2.0 * 1.0, rtol=1e-9)


def test_tile_partition_sums_to_full_frame(oracle):
    scene, cam, _ = rpt_amd.scenes.sphere_scene()
    osc = oracle.OracleScene(scene)
    full = osc.render(cam, make_params(40, 24, 2, 2, seed=5), threads=2)
    acc = np.zeros_like(full)
    for part in range(3):
        acc += osc.render(cam, make_params(40, 24, 2, 2, seed=5, tile=(8, 4), part=(part, 3)), threads=2)
    assert (acc == full).all()
    a = osc.render(cam, make_params(40, 24, 2, 2, seed=5), threads=1)
    assert (a == full).all()  # thread count does not change the image
