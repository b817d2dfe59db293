"""An independent path tracer, end to end.

tests/test_independent_shading.py and tests/test_independent_geometry.py restate the LEAVES of the path in numpy, from the
Rust text alone.  This file adds the GLUE — `Renderer::get_color`'s sample loop, `trace_ray`, `sample_lights`,
`get_closest_hit` (src/renderer.rs:131-220) and `Light::illuminate` (src/light.rs:23-47) — written from the same text,
composes it with those leaves (their functions are imported: bsdf, sample_f, the shape intersections and samples, the
camera ray, the HDRI lookup; numpy's libm, numpy's matrix inverses) and traces whole samples on the oracle's Philox
stream.  The radiance of every sample is then compared with the oracle's `trace_sample` for the same (pixel, sample):
`oracle.cpp` and the device kernels are twins, so a misreading of the recursion, of the order of the random draws
between light sampling and BSDF sampling, of the shadow test or of the object loop that both share would pass every
GPU-vs-oracle test — and fail here.

Two libms differ by an ulp now and then, and an ulp can flip a rejection test or a total-internal-reflection decision,
after which two correct tracers walk different paths: samples must agree to 1e-9 relative, and at most 1 % of them may
differ at all.  No GPU, no reference checkout at run time."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_ffi as O  # noqa: E402
from rpt_amd import (Camera, Environment, Light, Material, Object, Scene, _abi, cube, hex_color, make_params, plane,  # noqa: E402
                     scenes, sphere)
from rpt_amd import shape as S  # noqa: E402
import test_independent_geometry as G  # noqa: E402
import test_independent_shading as H  # noqa: E402

EPSILON = 1e-12         # renderer.rs:14
FIREFLY_CLAMP = 100.0   # renderer.rs:17


def v3(x):
    return np.array(x, dtype=float)


def dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def normalize(a):
    return a / math.sqrt(dot3(a, a))


# ------------------------------------------------------------------------------------------------ shapes (by composition)
LIB_MATRICES = False  # see scene_cornell


def intersect(shape, o, d, t_min, time):
    """Shape::intersect for the closed set used here -> (hit, time, normal); `time` is record.time on entry"""
    if isinstance(shape, S.Transformed):                                      # shape.rs:128-137
        if LIB_MATRICES:  # the host mirror's own inverse and inverse transpose (rpt_amd/glm.py: cofactors), column sums like nalgebra
            Minv = np.array(shape.inverse_transform).reshape(4, 4).T
            NT = np.array(shape.normal_transform).reshape(3, 3).T
            lo = np.array([((Minv[r, 0] * o[0] + Minv[r, 1] * o[1]) + Minv[r, 2] * o[2]) + Minv[r, 3] * 1.0 for r in range(3)])
            ld = np.array([((Minv[r, 0] * d[0] + Minv[r, 1] * d[1]) + Minv[r, 2] * d[2]) + Minv[r, 3] * 0.0 for r in range(3)])
        else:
            M = np.array(shape.transform_m).reshape(4, 4).T
            Minv = np.linalg.inv(M)
            NT = np.linalg.inv(M[:3, :3]).T
            lo = (Minv @ np.append(o, 1.0))[:3]
            ld = (Minv @ np.append(d, 0.0))[:3]
        hit, t, n = intersect(shape.shape, lo, ld, t_min, time)
        if hit:
            n = normalize(np.array([(NT[r, 0] * n[0] + NT[r, 1] * n[1]) + NT[r, 2] * n[2] for r in range(3)]))
        return hit, t, n
    if isinstance(shape, S.KdTree):                                           # a Mesh or a group: the Python kd-tree below
        return tree_of(shape).intersect(o, d, t_min, time)
    o1, d1 = o[None, :], d[None, :]
    tm, ti = np.array([t_min]), np.array([time])
    if isinstance(shape, S.Sphere):
        h, t, n = G.sphere_intersect(o1, d1, tm, ti)
    elif isinstance(shape, S.Cube):
        h, t, n = G.cube_intersect(o1, d1, tm, ti)
    elif isinstance(shape, S.Plane):
        h, t, n = G.plane_intersect(shape.normal, shape.value, o1, d1, tm, ti)
    else:
        raise TypeError(shape)
    return bool(h[0]), float(t[0]), n[0]


def sample_shape(shape, target, st):
    """Shape::sample -> (v, n, p)"""
    if isinstance(shape, S.Transformed):
        M = np.array(shape.transform_m).reshape(4, 4).T
        return G.transformed_sample(M, lambda tg, s: sample_shape(shape.shape, tg, s), target, st)
    if isinstance(shape, S.Sphere):
        return G.sphere_sample(target, st)
    if isinstance(shape, S.Cube):
        return G.cube_sample(target, st)
    if isinstance(shape, S.KdTree):
        return tree_of(shape).sample(target, st)
    raise TypeError(shape)


# ------------------------------------------------------------------------------------------------ KdTree, in Python
def fmin(a, b):
    """f64::min: a NaN operand is ignored"""
    return b if a != a else (a if b != b else min(a, b))


def fmax(a, b):
    return b if a != a else (a if b != b else max(a, b))


def div(a, b):
    """IEEE division (Python raises on a zero divisor)"""
    with np.errstate(all="ignore"):
        return float(np.float64(a) / np.float64(b))


class PyKdTree:
    """KdTree<Triangle> (src/kdtree.rs:100-355 with src/shape/mesh.rs:8-98), written from the Rust text: `construct` by
    the reference rule (three stable sorts per node, median of the box edges, the 0.85 score threshold, the
    longest-extent-first choice), `intersect_subtree` as the reference's recursion (the child boxes by
    BoundingBox::split, six divisions per visited node), the leaf's triangles in index order."""

    def __init__(self, rows=None, shapes=None):
        """rows: (n, 18) triangles — a Mesh; or shapes: bounded shapes — KdTree<Box<dyn Bounded>> (fractal_spheres.rs:45)"""
        self.shapes = shapes
        if shapes is not None:
            self.tri = list(shapes)  # (only its length is used below)
            boxes = [bounding_box(sh) for sh in shapes]
            self.lo, self.hi = [b[0] for b in boxes], [b[1] for b in boxes]
        else:
            r = np.asarray(rows, float)
            self.tri = [tuple(r[i, 3 * k:3 * k + 3].copy() for k in range(6)) for i in range(len(r))]
            self.lo = [np.minimum(np.minimum(t[0], t[1]), t[2]) for t in self.tri]  # Triangle::bounding_box (mesh.rs:40-45)
            self.hi = [np.maximum(np.maximum(t[0], t[1]), t[2]) for t in self.tri]
        pmin, pmax = np.full(3, math.inf), np.full(3, -math.inf)                      # KdTree::new (kdtree.rs:108-119)
        for a, b in zip(self.lo, self.hi):
            pmin, pmax = np.minimum(pmin, a), np.maximum(pmax, b)
        self.bounds = (pmin, pmax)
        self.root = self.construct(list(range(len(self.tri))))

    @staticmethod
    def median(a):                                                                     # kdtree.rs:347-355
        mid = len(a) // 2
        return a[mid] if len(a) % 2 else (a[mid] + a[mid - 1]) / 2.0

    def construct(self, indices):                                                      # kdtree.rs:235-345
        if len(indices) < 16:
            return ("leaf", indices)
        edges = [[], [], []]
        for i in indices:
            for k in range(3):
                edges[k].append(float(self.lo[i][k]))
                edges[k].append(float(self.hi[i][k]))
        med = [self.median(sorted(e)) for e in edges]                                  # (sorted() is stable and holds -0.0 == 0.0)

        def score(dim, value):
            left = sum(1 for i in indices if self.lo[i][dim] <= value)
            right = sum(1 for i in indices if self.hi[i][dim] >= value)
            return max(left, right)
        sc = [score(k, med[k]) for k in range(3)]
        threshold = int(len(indices) * 0.85)
        if min(sc) >= threshold:
            return ("leaf", indices)
        pmin, pmax = np.full(3, math.inf), np.full(3, -math.inf)
        for i in indices:
            pmin, pmax = np.minimum(pmin, self.lo[i]), np.maximum(pmax, self.hi[i])
        ext = pmax - pmin
        split = -1
        if ext[0] > ext[1] and ext[0] > ext[2]:
            if sc[0] < threshold:
                split = 0
        elif ext[1] > ext[2]:
            if sc[1] < threshold:
                split = 1
        elif sc[2] < threshold:
            split = 2
        if split == -1:
            split = 0 if (sc[0] < sc[1] and sc[0] < sc[2]) else (1 if sc[1] < sc[2] else 2)
        v = med[split]
        left = [i for i in indices if self.lo[i][split] <= v]
        right = [i for i in indices if self.hi[i][split] >= v]
        return (split, v, self.construct(left), self.construct(right))

    @staticmethod
    def box_times(lo, hi, o, d):                                                       # BoundingBox::intersect (kdtree.rs:53-69)
        mn, mx = -math.inf, math.inf
        first = True
        for k in range(3):
            a, b = div(lo[k] - o[k], d[k]), div(hi[k] - o[k], d[k])
            a, b = fmin(a, b), fmax(a, b)
            mn, mx = (a, b) if first else (fmax(mn, a), fmin(mx, b))
            first = False
        return mn, mx

    def tri_hit(self, i, o, d, t_min, rec):                                            # Triangle::intersect (mesh.rs:49-82)
        if self.shapes is not None:  # a group's child: its own Shape::intersect against the shared record
            h, t, n = intersect(self.shapes[i], o, d, t_min, rec[0])
            if h:
                rec[0], rec[1] = t, n
            return h
        v1, v2, v3, n1, n2, n3 = self.tri[i]
        d0, d1 = v2 - v1, v3 - v1
        pn = np.cross(d0, d1)
        pn = pn / math.sqrt(dot3(pn, pn))
        cosine = dot3(pn, d)
        if abs(cosine) < 1e-8:
            return False
        time = div(dot3(pn, v1 - o), cosine)
        if time < t_min or time >= rec[0]:
            return False
        d2 = (o + time * d) - v1
        d00, d01, d11 = dot3(d0, d0), dot3(d0, d1), dot3(d1, d1)
        d20, d21 = dot3(d2, d0), dot3(d2, d1)
        denom = d00 * d11 - d01 * d01
        v = div(d11 * d20 - d01 * d21, denom)
        w = div(d00 * d21 - d01 * d20, denom)
        u = 1.0 - v - w
        if u >= 0.0 and v >= 0.0 and w >= 0.0:
            rec[0] = time
            rec[1] = normalize(u * n1 + v * n2 + w * n3)
            return True
        return False

    def subtree(self, node, lo, hi, o, d, t_min, rec):                                 # intersect_subtree (kdtree.rs:150-223)
        b_min, b_max = self.box_times(lo, hi, o, d)
        if node[0] == "leaf":
            result = False
            for i in node[1]:
                if self.tri_hit(i, o, d, t_min, rec):
                    result = True
            return result
        ax, value, left, right = node
        t_split = div(value - o[ax], d[ax])
        left_first = (o[ax] < value) or (o[ax] == value and d[ax] <= 0.0)
        hi_l, lo_r = hi.copy(), lo.copy()
        hi_l[ax] = value
        lo_r[ax] = value
        if left_first:
            first, second = (left, lo, hi_l), (right, lo_r, hi)
        else:
            first, second = (right, lo_r, hi), (left, lo, hi_l)
        if t_split > fmin(b_max, rec[0]) or t_split <= 0.0:
            return self.subtree(first[0], first[1], first[2], o, d, t_min, rec)
        if t_split < fmax(b_min, t_min):
            return self.subtree(second[0], second[1], second[2], o, d, t_min, rec)
        h1 = self.subtree(first[0], first[1], first[2], o, d, t_min, rec)
        if h1 and rec[0] < t_split:
            return True
        h2 = self.subtree(second[0], second[1], second[2], o, d, t_split, rec)
        return h1 or h2

    def intersect(self, o, d, t_min, time):                                            # KdTree::intersect (kdtree.rs:129-136)
        b_min, b_max = self.box_times(self.bounds[0], self.bounds[1], o, d)
        if fmax(b_min, t_min) > fmin(b_max, time):
            return False, time, np.zeros(3)
        rec = [time, np.zeros(3)]
        hit = self.subtree(self.root, self.bounds[0].copy(), self.bounds[1].copy(), o, d, t_min, rec)
        return hit, rec[0], rec[1]

    def sample(self, target, st):                                                      # KdTree::sample (kdtree.rs:138-143)
        j = G.uniform_int(st, len(self.tri))
        v, n, p = sample_shape(self.shapes[j], target, st) if self.shapes is not None else G.triangle_sample(self.tri[j], st)
        return v, n, p / len(self.tri)

    def depth(self, node=None):
        node = node or self.root
        return 0 if node[0] == "leaf" else 1 + max(self.depth(node[2]), self.depth(node[3]))


def bounding_box(shape):
    """Bounded::bounding_box: Sphere (sphere.rs:67-74), Cube (cube.rs:10-17), Transformed<T> (shape.rs:153-176: the eight
    corners through the matrix, then componentwise min / max)"""
    if isinstance(shape, S.Sphere):
        return np.full(3, -1.0), np.full(3, 1.0)
    if isinstance(shape, S.Cube):
        return np.full(3, -0.5), np.full(3, 0.5)
    if isinstance(shape, S.Transformed):
        lo, hi = bounding_box(shape.shape)
        M = np.array(shape.transform_m).reshape(4, 4).T
        pts = []
        for x in (lo[0], hi[0]):
            for y in (lo[1], hi[1]):
                for z in (lo[2], hi[2]):
                    pts.append(np.array([((M[r, 0] * x + M[r, 1] * y) + M[r, 2] * z) + M[r, 3] * 1.0 for r in range(3)]))
        pts = np.array(pts)
        return pts.min(axis=0), pts.max(axis=0)
    raise TypeError(shape)


def tree_of(shape):
    """the Python kd-tree of a Mesh or a group, built once and kept ON the shape (an id()-keyed cache would outlive it)"""
    if getattr(shape, "_py_tree", None) is None:
        shape._py_tree = PyKdTree(rows=shape.triangles) if shape.triangles is not None else PyKdTree(shapes=shape.objects)
    return shape._py_tree


# ------------------------------------------------------------------------------------------------ the glue
class Tracer:
    def __init__(self, scene, camera, width, height, max_bounces):
        self.scene, self.camera, self.w, self.h, self.max_bounces = scene, camera, width, height, max_bounces
        env = scene.environment
        self.hdri = None if env.hdri is None else np.asarray(env.hdri.buf, float).reshape(env.hdri.height, env.hdri.width, 3)

    def closest_hit(self, o, d):
        """get_closest_hit (renderer.rs:211-220): every object in turn against ONE record"""
        time, normal, hit = math.inf, np.zeros(3), None
        for obj in self.scene.objects:
            h, t, n = intersect(obj.shape, o, d, EPSILON, time)
            if h:
                time, normal, hit = t, n, obj
        return (time, normal, hit) if hit is not None else None

    def illuminate(self, light, pos, st):
        """Light::illuminate (light.rs:23-47) -> (intensity, wi, dist)"""
        color = v3(light.color)
        if light.kind == _abi.RPT_LIGHT_POINT:
            disp = v3(light.vec) - pos
            ln = math.sqrt(dot3(disp, disp))
            return color / (ln * ln), disp / ln, ln
        if light.kind == _abi.RPT_LIGHT_DIRECTIONAL:
            return color, -normalize(v3(light.vec)), math.inf
        obj = light.object
        v, n, p = sample_shape(obj.shape, pos, st)
        disp = v - pos
        ln = math.sqrt(dot3(disp, disp))
        cosine = max(-dot3(disp, n), 0.0) / ln
        area = max(cosine, 0.0) / (ln * ln)
        m = obj._material
        return v3(m.color) * m.emittance * area / p, disp / ln, ln

    def bsdf(self, m, n, wo, wi):
        arr = {"color": v3(m.color)[None, :], "index": np.array([m.index]), "roughness": np.array([m.roughness]),
               "metallic": np.array([m.metallic]), "transparent": np.array([bool(m.transparent)])}
        return H.bsdf(arr, n[None, :], wo[None, :], wi[None, :])[0]

    def sample_lights(self, m, pos, n, wo, st):
        """renderer.rs:177-204"""
        color = np.zeros(3)
        for light in self.scene.lights:
            if light.kind == _abi.RPT_LIGHT_AMBIENT:
                color = color + v3(light.color) * v3(m.color)
                continue
            intensity, wi, dist = self.illuminate(light, pos, st)
            hit = self.closest_hit(pos, wi)
            if hit is None or hit[0] > dist:
                f = self.bsdf(m, n, wo, wi)
                color = color + f * intensity * dot3(wi, n)
        return color

    def env(self, d):
        e = self.scene.environment
        return v3(e.color) if self.hdri is None else G.hdri_color(self.hdri, d)

    def trace_ray(self, o, d, bounces, st):
        """renderer.rs:145-174"""
        hit = self.closest_hit(o, d)
        if hit is None:
            return self.env(d)
        time, n, obj = hit
        world_pos = o + time * d
        m = obj._material
        wo = -normalize(d)
        color = m.emittance * v3(m.color)
        color = color + self.sample_lights(m, world_pos, n, wo, st)
        if bounces < self.max_bounces:
            r = H.sample_f(m, n, wo, st)
            if r is not None:
                wi, pdf = r
                f = self.bsdf(m, n, wo, wi)
                indirect = 1.0 / pdf * (f * self.trace_ray(world_pos, wi, bounces + 1, st)) * abs(dot3(wi, n))
                color = color + np.minimum(indirect, FIREFLY_CLAMP)
        return color

    def sample(self, seed, x, y, s):
        st = H.Stream(seed, y * self.w + x, s, 0)
        o, d = G.camera_ray(self.camera, self.w, self.h, x, y, st)
        return self.trace_ray(o, d, 0, st)


# ------------------------------------------------------------------------------------------------ scenes
def scene_lights_and_shapes():
    """every light kind, every primitive, transformed and not, diffuse / specular / metallic / glass"""
    sc = Scene()
    sc.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    sc.add(Object(sphere().translate((-1.3, 0.0, 0.0))).material(Material.specular(hex_color(0x4060C0), 0.3)))
    sc.add(Object(sphere().scale((0.6, 0.9, 0.6)).translate((1.4, -0.1, 0.4))).material(Material.clear(1.5, 0.05)))
    sc.add(Object(cube().rotate_y(0.5).scale((0.9, 1.4, 0.9)).translate((0.1, -0.3, -1.6))).material(Material.metallic_(hex_color(0xE0B060), 0.2)))
    sc.add(Object(cube().scale((0.5, 0.5, 0.5)).translate((0.2, -0.75, 1.3))).material(Material.transparent_((0.7, 1.0, 0.8), 1.3, 0.2)))
    # (the lamps are lights only, not visible objects: a shadow ray towards a point ON a visible lamp hits the lamp at the
    # light's distance give or take an ulp, and `closest_hit > dist_to_light` (renderer.rs:197) then depends on the last
    # bit of the transformed ray — numpy's inverse here, cofactors in the library: two correct tracers disagree on 5 % of
    # the samples of such a scene)
    sc.add(Light.Object(Object(sphere().scale((0.4, 0.4, 0.4)).translate((0.5, 3.0, 1.5))).material(Material.light((1.0, 0.9, 0.8), 40.0))))
    sc.add(Light.Object(Object(cube().scale((0.6, 0.1, 0.6)).translate((-2.0, 2.5, 0.5))).material(Material.light((0.8, 0.9, 1.0), 30.0))))
    sc.add(Light.Ambient((0.03, 0.03, 0.04)))
    sc.add(Light.Point((30.0, 28.0, 25.0), (3.0, 4.0, 3.0)))
    sc.add(Light.Directional((0.4, 0.4, 0.5), (0.3, -1.0, -0.2)))
    return sc, Camera.look_at((0.5, 1.6, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.7)


def scene_glass_and_sky():
    """examples/glass.rs in small: a mirror ball and a glass ball under an HDRI, no lights, depth of field"""
    sc = Scene()
    sc.environment = Environment.Hdri(scenes.synthetic_hdri(64, 32, seed=4))
    sc.add(Object(sphere().translate((1.1, 0.0, 0.0))).material(Material.metallic_(hex_color(0xFFFFFF), 0.05)))
    sc.add(Object(sphere().translate((-1.1, 0.0, 0.0))).material(Material.clear(1.5, 0.05)))
    return sc, Camera.look_at((0.0, 0.5, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.6).focus((0.0, 0.0, 0.0), 0.05)


def scene_cornell():
    """the headline's scene itself (examples/cornell.rs): five quad walls, two rotated boxes, the quad light.  Compared
    with LIB_MATRICES: a ray leaving a face of a box is tested against that box again, and whether it re-hits its own face
    is decided by the last bit of the transformed ray (renderer.rs:14: t_min = 1e-12 is all that separates them) — with
    numpy's inverse instead of the host mirror's, 1.5 % of the samples take the other side.  The inverse itself is held
    against numpy's in tests/test_independent_geometry.py."""
    sc, cam, _ = scenes.cornell()
    return sc, cam


def scene_mesh():
    """a 1536-triangle mesh with a real kd-tree (the Python builder's, by the reference rule) over a plane, a sphere light
    and a point light: `construct`, `intersect_subtree` and the leaf loop inside whole paths"""
    sc = Scene()
    rows = scenes.knot_mesh(64, 12, seed=0x7E57)
    rows[:, :9] *= 1.6
    mesh = S.Mesh(rows)
    assert tree_of(mesh).depth() >= 5
    sc.add(Object(mesh).material(Material.specular(hex_color(0xB7CA79), 0.25)))
    sc.add(Object(plane((0.0, 1.0, 0.0), -1.6)).material(Material.diffuse(hex_color(0x999999))))
    sc.add(Object(plane((0.0, 0.0, 1.0), -3.0)).material(Material.diffuse(hex_color(0xBB8866))))   # a back wall and a side wall keep the paths going
    sc.add(Object(plane((1.0, 0.0, 0.0), -3.5)).material(Material.specular(hex_color(0x6688BB), 0.4)))
    sc.add(Light.Object(Object(sphere().scale((0.5, 0.5, 0.5)).translate((0.0, 5.0, 2.0))).material(Material.light((1.0, 1.0, 0.9), 60.0))))
    sc.add(Light.Point((20.0, 20.0, 24.0), (-3.0, 3.0, 4.0)))
    sc.add(Light.Ambient((0.02, 0.02, 0.02)))
    return sc, Camera.look_at((0.5, 1.2, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.7)


def scene_fractal_spheres():
    """examples/fractal_spheres.rs at three levels: groups of 1, 5 and 25 placed spheres — the last one a real
    KdTree<Box<dyn Bounded>> — over a plane, with the example's lights"""
    sc, cam, _ = scenes.fractal_spheres(levels=3)
    assert tree_of(sc.objects[2].shape).depth() >= 1
    return sc, cam


def compare(make, width, height, bounces, n_samples, seed):
    sc, cam = make()
    tr = Tracer(sc, cam, width, height, bounces)
    p = make_params(width, height, bounces, 1, seed=seed)
    osc = O.OracleScene(sc)
    rs = np.random.RandomState(seed)
    same = close = deep = 0
    for _ in range(n_samples):
        x, y, s = int(rs.randint(0, width)), int(rs.randint(0, height)), int(rs.randint(0, 64))
        want, rec = osc.trace_sample(cam, p, x, y, s)
        got = tr.sample(seed, x, y, s)
        deep += len(rec) > 2
        tol = 1e-9 * max(1.0, float(np.abs(want).max()))
        if np.abs(got - want).max() <= tol:
            close += 1
            same += int((got == want).all())
    return close / n_samples, same / n_samples, deep / n_samples


def test_whole_samples_agree_with_an_independent_path_tracer():
    frac, same, deep = compare(scene_lights_and_shapes, 64, 48, 4, 1500, 4242)
    assert deep > 0.3                      # paths that bounce at least twice: the recursion and the clamp's fold are exercised
    assert frac >= 0.99, (frac, same)      # measured: 1500 of 1500 agree to 1e-9, 62 % bit for bit, 34 % of the paths bounce twice or more
    frac, same, deep = compare(scene_glass_and_sky, 48, 36, 6, 1200, 99)
    assert deep > 0.1
    assert frac >= 0.99, (frac, same)      # measured: 1200 of 1200, 50 % bit for bit


def test_the_cornell_box_and_a_kd_tree_mesh_agree_with_the_independent_path_tracer():
    """the same with triangles: the Python KdTree (construct + intersect_subtree + Triangle::intersect / sample) inside the
    independent tracer — on the headline's own scene, and on a mesh whose tree is five levels deep and more"""
    global LIB_MATRICES
    LIB_MATRICES = True
    try:
        frac, same, deep = compare(scene_cornell, 64, 36, 8, 600, 512)
    finally:
        LIB_MATRICES = False
    assert deep > 0.3 and frac >= 0.99, (frac, same, deep)   # measured: 598 of 600 to 1e-9, 84 % bit for bit, 68 % of the paths three vertices and more
    frac, same, deep = compare(scene_mesh, 48, 36, 4, 500, 77)
    assert deep > 0.2 and frac >= 0.99, (frac, same, deep)
    frac, same, deep = compare(scene_fractal_spheres, 64, 36, 6, 500, 31)
    assert deep > 0.2 and frac >= 0.99, (frac, same, deep)


def test_python_kd_builder_equals_the_product_builder_node_for_node():
    """a fourth builder: PyKdTree.construct (sorted(), median, scores — from the Rust text) against rptgpu_kdtree_build
    (order statistics, threads) on two meshes and on the mixed-sign-zero boxes of tests/test_kdtree.py: the same axis,
    the same split BITS, the same leaf entries in the same order"""
    from rpt_amd.device import kdtree_build
    import small_scenes

    def walk(py, t, idx, count):
        count[0] += 1
        if py[0] == "leaf":
            assert t["info"][idx] & 3 == 3
            first, n = int(t["a"][idx]), int(t["b"][idx])
            assert list(t["refs"][first:first + n]) == list(py[1]), idx
            return
        ax, value, left, right = py
        assert int(t["info"][idx]) == ax, idx
        assert np.float64(value).view(np.uint64) == t["split"][idx:idx + 1].view(np.uint64)[0], (idx, value, t["split"][idx])
        walk(left, t, int(t["a"][idx]), count)
        walk(right, t, int(t["a"][idx]) + 1, count)

    cases = [scenes.knot_mesh(64, 12, seed=0x7E57), scenes.lathe_glass_mesh(24)] + \
            [small_scenes.mixed_zero_mesh(order) for order, _ in small_scenes.MIXED_ZERO_ORDERS]
    for rows in cases:
        py = PyKdTree(rows=rows)
        v = np.asarray(rows)[:, :9].reshape(-1, 3, 3)
        t = kdtree_build(np.concatenate([v.min(axis=1), v.max(axis=1)], axis=1))
        count = [0]
        walk(py.root, t, 0, count)
        assert count[0] == len(t["split"])
