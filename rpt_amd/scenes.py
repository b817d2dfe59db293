"""The benchmark scenes of BASELINE.json, written against the rpt builder API exactly as the
reference's example programs are (reference examples/*.rs, cited per function).  Assets the
reference downloads (dragon.obj, the ballroom HDRIs) or reads from its own tree
(wine_glass.obj) are replaced by seeded procedural stand-ins generated here.

Each function returns (scene, camera, defaults) where defaults carries the BASELINE config:
width, height, max_bounces, num_samples.
"""
import math

import numpy as np

from .camera import Camera
from .color import hex_color
from .environment import Environment, Hdri
from .light import Light
from .material import Material
from .object import Object
from .scene import Scene
from .shape import KdTree, Mesh, cube, monomial_surface, plane, polygon, sphere


# ----------------------------------------------------------------------------- C1
def sphere_scene():
    """examples/sphere.rs:3-35 — 960x540, 2 bounces, 100 spp."""
    scene = Scene()
    scene.add(Object(sphere()))  # default red material
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Object(
        Object(sphere().scale((2.0, 2.0, 2.0)).translate((0.0, 12.0, 0.0)))
        .material(Material.light(hex_color(0xFFFFFF), 40.0))))
    camera = Camera.look_at((-2.5, 4.0, 6.5), (0.0, -0.25, 0.0), (0.0, 1.0, 0.0), math.pi / 4.0)
    return scene, camera, dict(width=960, height=540, max_bounces=2, num_samples=100)


# ----------------------------------------------------------------------------- C2
def cornell():
    """examples/cornell.rs:9-80 — BASELINE: 1920x1080, 8 bounces, 512 spp."""
    scene = Scene()
    camera = Camera(eye=(278.0, 273.0, -800.0), direction=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0), fov=0.686)
    white = Material.diffuse(hex_color(0xAAAAAA))
    red = Material.diffuse(hex_color(0xBC0000))
    green = Material.diffuse(hex_color(0x00BC00))
    light_mtl = Material.light(hex_color(0xFFFEFA), 100.0)
    floor = polygon([(0.0, 0.0, 0.0), (0.0, 0.0, 559.2), (556.0, 0.0, 559.2), (556.0, 0.0, 0.0)])
    ceiling = polygon([(0.0, 548.9, 0.0), (556.0, 548.9, 0.0), (556.0, 548.9, 559.2), (0.0, 548.9, 559.2)])
    light_rect = polygon([(343.0, 548.8, 227.0), (343.0, 548.8, 332.0), (213.0, 548.8, 332.0), (213.0, 548.8, 227.0)])
    back_wall = polygon([(0.0, 0.0, 559.2), (0.0, 548.9, 559.2), (556.0, 548.9, 559.2), (556.0, 0.0, 559.2)])
    right_wall = polygon([(0.0, 0.0, 0.0), (0.0, 548.9, 0.0), (0.0, 548.9, 559.2), (0.0, 0.0, 559.2)])
    left_wall = polygon([(556.0, 0.0, 0.0), (556.0, 0.0, 559.2), (556.0, 548.9, 559.2), (556.0, 548.9, 0.0)])
    two_pi = 2.0 * math.pi
    large_box = (cube().scale((165.0, 330.0, 165.0)).rotate_y(two_pi * (-253.0 / 360.0))
                 .translate((368.0, 165.0, 351.0)))
    small_box = (cube().scale((165.0, 165.0, 165.0)).rotate_y(two_pi * (-197.0 / 360.0))
                 .translate((185.0, 82.5, 169.0)))
    scene.add(Object(floor).material(white))
    scene.add(Object(ceiling).material(white))
    scene.add(Object(back_wall).material(white))
    scene.add(Object(left_wall).material(red))
    scene.add(Object(right_wall).material(green))
    scene.add(Object(large_box).material(white))
    scene.add(Object(small_box).material(white))
    scene.add(Light.Object(Object(light_rect).material(light_mtl)))
    return scene, camera, dict(width=1920, height=1080, max_bounces=8, num_samples=512)


# ----------------------------------------------------------------------------- meshes
def _smooth_mesh(verts, faces):
    """(V,3) vertices + (F,3) indices -> (F,18) triangle rows with area-weighted vertex normals."""
    v = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    # closed surfaces: orient outwards (positive signed volume) so normals leave the solid
    if float((a * np.cross(b, c)).sum()) < 0.0:
        f = f[:, [0, 2, 1]]
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    fn = np.cross(b - a, c - a)
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, f[:, k], fn)
    ln = np.sqrt((vn * vn).sum(axis=1, keepdims=True))
    vn = vn / np.where(ln > 0, ln, 1.0)
    return np.ascontiguousarray(np.concatenate([a, b, c, vn[f[:, 0]], vn[f[:, 1]], vn[f[:, 2]]], axis=1))


def knot_mesh(nu=784, nv=64, seed=0x5EED):
    """The dragon stand-in: a displaced (2,3) torus-knot tube, nu*nv*2 triangles (default
    100 352), smooth normals, fitted into a unit-extent box resting at y = -1/3.4 so that the
    dragon example's `.scale(3.4)` puts it on the floor plane y = -1."""
    rs = np.random.RandomState(seed)
    u = np.arange(nu) * (2.0 * math.pi / nu)
    p, q = 2.0, 3.0
    r = 2.0 + np.cos(q * u)
    center = np.stack([r * np.cos(p * u), np.sin(q * u) * 1.2, r * np.sin(p * u)], axis=1)
    tangent = np.roll(center, -1, axis=0) - np.roll(center, 1, axis=0)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(tangent, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(tangent, n1)
    v = np.arange(nv) * (2.0 * math.pi / nv)
    # smooth seeded displacement: a few low-frequency harmonics in (u, v)
    amp = rs.rand(6) * 0.08
    fu = rs.randint(3, 17, size=6)
    fv = rs.randint(1, 5, size=6)
    ph = rs.rand(6) * 2.0 * math.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    radius = 0.42 + sum(amp[k] * np.sin(fu[k] * uu + fv[k] * vv + ph[k]) for k in range(6))
    pts = (center[:, None, :] + radius[:, :, None] * (np.cos(vv)[:, :, None] * n1[:, None, :]
                                                        + np.sin(vv)[:, :, None] * n2[:, None, :]))
    verts = pts.reshape(-1, 3)
    lo, hi = verts.min(axis=0), verts.max(axis=0)
    verts = (verts - (lo + hi) / 2.0) / (hi - lo).max()
    verts[:, 1] += -1.0 / 3.4 - verts[:, 1].min()
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i00 = (iu * nv + iv).ravel()
    i10 = (((iu + 1) % nu) * nv + iv).ravel()
    i01 = (iu * nv + (iv + 1) % nv).ravel()
    i11 = (((iu + 1) % nu) * nv + (iv + 1) % nv).ravel()
    faces = np.concatenate([np.stack([i00, i10, i11], axis=1), np.stack([i00, i11, i01], axis=1)])
    return _smooth_mesh(verts, faces)


def lathe_glass_mesh(segments=128):
    """The wine-glass stand-in: a two-sided lathed profile (outer wall up, inner wall down) so
    rays refract in and out; about 16k triangles at 128 segments, y-up, foot on y = 0."""
    # (radius, height) going up the outside, then down the inside of the bowl
    outer = [(0.0, 0.0), (0.9, 0.0), (0.95, 0.04), (0.9, 0.08), (0.3, 0.14), (0.12, 0.25), (0.09, 0.6),
             (0.09, 1.4), (0.12, 1.75), (0.3, 1.95), (0.62, 2.2), (0.86, 2.55), (0.98, 2.95),
             (1.02, 3.4), (1.0, 3.85), (0.93, 4.25), (0.86, 4.6)]
    inner = [(0.83, 4.6), (0.9, 4.25), (0.97, 3.85), (0.99, 3.4), (0.95, 2.97), (0.83, 2.59),
             (0.6, 2.26), (0.3, 2.03), (0.0, 1.97)]

    def refine(pts, times):
        for _ in range(times):
            out = [pts[0]]
            for a, b in zip(pts[:-1], pts[1:]):
                out += [((a[0] + b[0]) / 2.0, (a[1] + b[1]) / 2.0), b]
            pts = out
        return pts

    prof = np.array(refine(outer + inner, 1), dtype=np.float64)
    prof[:, 1] *= 0.55  # overall height ~2.5 like the reference's glass in a 10x10 quad
    prof[:, 0] *= 0.55
    ang = np.arange(segments) * (2.0 * math.pi / segments)
    n = len(prof)
    verts = np.stack([prof[:, None, 0] * np.cos(ang)[None, :],
                      np.repeat(prof[:, 1:2], segments, axis=1),
                      prof[:, None, 0] * np.sin(ang)[None, :]], axis=2).reshape(-1, 3)
    faces = []
    for i in range(n - 1):
        for j in range(segments):
            j2 = (j + 1) % segments
            a, b, c, d = i * segments + j, i * segments + j2, (i + 1) * segments + j2, (i + 1) * segments + j
            if prof[i, 0] > 0.0:
                faces.append((a, c, b))
            if prof[i + 1, 0] > 0.0:
                faces.append((a, d, c))
    return _smooth_mesh(verts, np.array(faces))


def synthetic_hdri(width=2048, height=1024, seed=0xBA11):
    """Stand-in for ballroom_*.hdr: sky/ground gradient + 8 Gaussian lamps (radiance <= 50)
    + low-amplitude value noise.  f32-rounded like a decoded .hdr (examples/glass.rs:11-13)."""
    rs = np.random.RandomState(seed)
    y = (np.arange(height) + 0.5) / height
    x = (np.arange(width) + 0.5) / width
    yy, xx = np.meshgrid(y, x, indexing="ij")
    sky = np.stack([0.35 + 0.4 * (1 - yy), 0.4 + 0.45 * (1 - yy), 0.55 + 0.6 * (1 - yy)], axis=2)
    ground = np.stack([0.25 * yy, 0.2 * yy, 0.15 * yy], axis=2)
    img = np.where(yy[:, :, None] < 0.55, sky, ground + 0.08)
    for _ in range(8):
        cx, cy = rs.rand(), 0.08 + 0.4 * rs.rand()
        sig = 0.006 + 0.02 * rs.rand()
        power = 8.0 + 42.0 * rs.rand()
        tint = 0.8 + 0.2 * rs.rand(3)
        dx = np.minimum(np.abs(xx - cx), 1.0 - np.abs(xx - cx))
        g = np.exp(-(dx * dx + (yy - cy) ** 2) / (2.0 * sig * sig))
        img = img + power * g[:, :, None] * tint[None, None, :]
    coarse = rs.rand(height // 32 + 2, width // 32 + 2, 3)
    noise = np.kron(coarse, np.ones((32, 32, 1)))[:height, :width]
    img = img * (0.95 + 0.1 * noise)
    return Hdri(width, height, img.astype(np.float32).astype(np.float64))


# ----------------------------------------------------------------------------- C3
def dragon(nu=784, nv=64, shape=None):
    """examples/dragon.rs:30-71 with the knot stand-in for dragon.obj —
    BASELINE: 1920x1080, 8 bounces, 256 spp.  `shape`: an already placed shape to use instead of the stand-in
    (SURVEY §8d names pegasus.obj, scaled x2 and dropped onto y = -1, as the alternative)."""
    scene = Scene()
    if shape is None:
        shape = Mesh(knot_mesh(nu, nv)).scale((3.4, 3.4, 3.4)).rotate_y(math.pi / 2.0)
    scene.add(Object(shape).material(Material.specular(hex_color(0xB7CA79), 0.1)))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Ambient((0.01, 0.01, 0.01)))
    scene.add(Light.Object(Object(sphere().scale((2.0, 2.0, 2.0)).translate((0.0, 20.0, 3.0)))
                           .material(Material.light((1.0, 1.0, 1.0), 160.0))))
    scene.add(Light.Object(Object(sphere().scale((0.05, 0.05, 0.05)).translate((-1.0, 0.71, 0.0)))
                           .material(Material.light(hex_color(0xFFAAAA), 400.0))))
    camera = Camera.look_at((-2.5, 4.0, 6.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), math.pi / 6.0)
    return scene, camera, dict(width=1920, height=1080, max_bounces=8, num_samples=256)


# ----------------------------------------------------------------------------- C4
def fractal_spheres(levels=5):
    """examples/fractal_spheres.rs:3-76 — BASELINE: 3840x2160, 8 bounces, 1024 spp."""
    colors = [0x264653, 0x2A9D8F, 0xE9C46A, 0xF4A261, 0xE76F51][:levels]
    spheres = [[] for _ in colors]

    def gen(p, rad, depth, last_dir):  # fractal_spheres.rs:3-31
        spheres[depth].append(sphere().scale((rad, rad, rad)).translate(p))
        if depth == len(spheres) - 1:
            return
        disp = rad * 7.0 / 5.0
        dx = [disp, -disp, 0.0, 0.0, 0.0, 0.0]
        dy = [0.0, 0.0, disp, -disp, 0.0, 0.0]
        dz = [0.0, 0.0, 0.0, 0.0, disp, -disp]
        for i in range(6):
            if last_dir is None or i != (last_dir ^ 1):
                gen((p[0] + dx[i], p[1] + dy[i], p[2] + dz[i]), rad * 2.0 / 5.0, depth + 1, i)

    gen((0.0, 0.0, 0.0), 1.0, 0, None)
    scene = Scene()
    for i, group in enumerate(spheres):
        scene.add(Object(KdTree(group)).material(Material.specular(hex_color(colors[i]), 0.25)))
    scene.add(Object(plane((0.0, 0.0, 1.0), -6.0)).material(Material.diffuse(hex_color(0xFFCCCC))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    n = math.sqrt((0.0 * 0.0 + 0.65 * 0.65) + 1.0 * 1.0)
    scene.add(Light.Directional((0.6, 0.6, 0.6), (0.0 / n, -0.65 / n, -1.0 / n)))
    scene.add(Light.Point((100.0, 100.0, 100.0), (0.0, 5.0, 5.0)))
    dn = math.sqrt((0.285714 * 0.285714 + 0.5 * 0.5) + 1.0 * 1.0)
    un = math.sqrt((0.0 + 1.0) + 0.25)
    camera = Camera(eye=(2.0, 3.5, 7.0), direction=(-0.285714 / dn, -0.5 / dn, -1.0 / dn),
                    up=(0.0 / un, 1.0 / un, -0.5 / un), fov=math.pi / 6.0)
    return scene, camera, dict(width=3840, height=2160, max_bounces=8, num_samples=1024)


# ----------------------------------------------------------------------------- C5
def _teapot_stand_in(nu, nv):
    rows = knot_mesh(nu, nv, seed=0x7EA)
    rows[:, :9] *= 3.0  # about the extent of teapot.obj (the example halves it again)
    return rows


def fractal_teapots(levels=5, mesh=None):
    """examples/fractal_teapots.rs:13-87 — a kd-tree of kd-trees: every level is a
    KdTree<Box<dyn Bounded>> whose children are Transformed<Arc<Mesh>> sharing ONE mesh.
    `mesh` is a Mesh (e.g. load_obj("examples/teapot.obj")); the default is a seeded procedural
    stand-in because the reference's asset is not redistributed here."""
    if mesh is None:
        mesh = Mesh(_teapot_stand_in(48, 8))
    colors = [0x264653, 0x2A9D8F, 0xE9C46A, 0xF4A261, 0xE76F51][:levels]
    groups = [[] for _ in colors]

    def gen(p, rad, depth, last_dir):  # fractal_teapots.rs:13-46
        groups[depth].append(mesh.scale((0.5, 0.5, 0.5)).scale((rad, rad, rad)).translate(p))
        if depth == len(groups) - 1:
            return
        disp = rad * 7.0 / 5.0
        dx = [disp, -disp, 0.0, 0.0, 0.0, 0.0]
        dy = [0.0, 0.0, disp, -disp, 0.0, 0.0]
        dz = [0.0, 0.0, 0.0, 0.0, disp, -disp]
        for i in range(6):
            if last_dir is None or i != (last_dir ^ 1):
                gen((p[0] + dx[i], p[1] + dy[i], p[2] + dz[i]), rad * 2.0 / 5.0, depth + 1, i)

    gen((0.0, 0.0, 0.0), 1.0, 0, None)
    scene = Scene()
    for i, group in enumerate(groups):
        scene.add(Object(KdTree(group)).material(Material.specular(hex_color(colors[i]), 0.25)))
    scene.add(Object(plane((0.0, 0.0, 1.0), -6.0)).material(Material.diffuse(hex_color(0xFFCCCC))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    n = math.sqrt((0.0 * 0.0 + 0.65 * 0.65) + 1.0 * 1.0)
    scene.add(Light.Directional((0.6, 0.6, 0.6), (0.0 / n, -0.65 / n, -1.0 / n)))
    scene.add(Light.Point((100.0, 100.0, 100.0), (0.0, 5.0, 5.0)))
    dn = math.sqrt((0.285714 * 0.285714 + 0.5 * 0.5) + 1.0 * 1.0)
    un = math.sqrt((0.0 + 1.0) + 0.25)
    camera = Camera(eye=(2.0, 3.5, 7.0), direction=(-0.285714 / dn, -0.5 / dn, -1.0 / dn),
                    up=(0.0 / un, 1.0 / un, -0.5 / un), fov=math.pi / 6.0)
    return scene, camera, dict(width=800, height=600, max_bounces=0, num_samples=1)


def glass(hdri_size=(2048, 1024)):
    """examples/glass.rs:27-50 (metal + glass unit spheres under an HDRI)."""
    scene = Scene()
    scene.environment = Environment.Hdri(synthetic_hdri(*hdri_size))
    scene.add(Object(sphere().translate((1.1, 0.0, 0.0))).material(Material.metallic_(hex_color(0xFFFFFF), 0.0001)))
    scene.add(Object(sphere().translate((-1.1, 0.0, 0.0))).material(Material.clear(1.5, 0.0001)))
    return scene, Camera(), dict(width=3840, height=2160, max_bounces=16, num_samples=4096)


def wine_glass(hdri_size=(2048, 1024), segments=128, mesh=None):
    """examples/wine_glass.rs:27-85 with the lathed stand-in for wine_glass.obj (or `mesh`, e.g. the loaded asset) —
    BASELINE: 3840x2160, 16 bounces, 4096 spp."""
    scene = Scene()
    scene.environment = Environment.Hdri(synthetic_hdri(*hdri_size))
    scene.add(Object(mesh if mesh is not None else Mesh(lathe_glass_mesh(segments))).material(Material.clear(1.5, 0.0001)))
    scene.add(Object(polygon([(-5.0, 0.0, -5.0), (-5.0, 0.0, 5.0), (5.0, 0.0, 5.0), (5.0, 0.0, -5.0)]))
              .material(Material.diffuse(hex_color(0x6F5D48))))
    scene.add(Light.Object(Object(sphere().scale((3.0, 3.0, 3.0)).translate((11.15, 13.739, -4.9325)))
                           .material(Material.light(hex_color(0xFFFFFF), 200.0))))
    eye = (5.530, 4.375, 5.384)
    camera = Camera.look_at(eye, (eye[0] - 0.6962, eye[1] - 0.3754, eye[2] - 0.6119), (0.0, 1.0, 0.0), 0.6911)
    return scene, camera, dict(width=3840, height=2160, max_bounces=16, num_samples=4096)


def _basic_objects(scene):
    """The cube / two spheres / floor / lights shared by examples/basic.rs:9-41 and monomial_glass.rs:36-71."""
    scene.add(Object(cube().rotate_y(math.pi / 6.0).scale((0.5, 0.3, 0.4)).translate((0.4, -0.8, 4.0)))
              .material(Material.specular(hex_color(0xFF00FF), 0.5)))
    scene.add(Object(sphere().scale((0.5, 0.5, 0.5)).translate((1.5, -0.5, 1.0)))
              .material(Material.specular(hex_color(0x0000FF), 0.1)))
    scene.add(Object(sphere().scale((0.5, 0.5, 0.5)).translate((-1.5, -0.5, 1.0)))
              .material(Material.specular(hex_color(0x00FF00), 0.1)))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.specular(hex_color(0xAAAAAA), 0.5)))
    scene.add(Light.Ambient((0.01, 0.01, 0.01)))
    scene.add(Light.Point((100.0, 100.0, 100.0), (0.0, 5.0, 5.0)))


def basic():
    """examples/basic.rs:6-50 (default material sphere, point + ambient light, default renderer settings)."""
    scene = Scene()
    scene.add(Object(sphere()))
    _basic_objects(scene)
    return scene, Camera(), dict(width=800, height=600, max_bounces=0, num_samples=1)


def monomial_glass(hdri_size=(2048, 1024)):
    """examples/monomial_glass.rs:26-87 (a mirror-like MonomialSurface bowl under an HDRI)."""
    scene = Scene()
    scene.environment = Environment.Hdri(synthetic_hdri(*hdri_size))
    scene.add(Object(monomial_surface(2.0, 4.0).translate((0.0, -1.0, 0.0)))
              .material(Material.metallic_(hex_color(0xFFFFFF), 0.0001)))
    _basic_objects(scene)
    return scene, Camera(), dict(width=800, height=600, max_bounces=1, num_samples=100)


def spheres():
    """examples/spheres.rs:13-60 (depth of field over five glossy spheres; Z-up)."""
    scene = Scene()
    balls = [((0.5, 4.0, 1.0), 0xE78999), ((3.15, -0.7, 1.5), 0xE7A94D), ((0.1, -2.0, 0.6), 0xB3E7AA),
             ((-1.7, -0.2, 1.1), 0x7CA3E7), ((1.2, 0.4, 0.5), 0xAAAAAA)]
    scene.add(Object(plane((0.0, 0.0, 1.0), 0.0)).material(Material.diffuse(hex_color(0xE7E7E7))))
    for pos, col in balls:
        scene.add(Object(sphere().scale((pos[2], pos[2], pos[2])).translate(pos))
                  .material(Material.specular(hex_color(col), 0.1)))
    scene.add(Light.Object(Object(sphere().scale((2.0, 2.0, 2.0)).translate((1.2, -1.5, 8.0)))
                           .material(Material.light(hex_color(0xFFFFFF), 8.0))))
    camera = Camera.look_at((0.7166, -9.2992, 2.8803), (0.8673, 0.2095, 0.9557), (0.0, 0.0, 1.0), 0.6911) \
        .focus((0.1, -2.0, 0.6), 0.15)
    return scene, camera, dict(width=800, height=600, max_bounces=6, num_samples=1000)


def compound():
    """examples/compound.rs:19-60 (compound of five cubes, three sphere lamps)."""
    scene = Scene()
    magic_angle = math.acos((3.0 * math.sqrt(5.0) - 1.0) / 8.0)
    c_central = cube()
    c_green = c_central.rotate(-magic_angle, (1.0, 1.0, 1.0))
    c_red = c_green.scale((-1.0, 1.0, 1.0))
    c_blue = c_green.scale((1.0, -1.0, 1.0))
    c_orange = c_red.scale((1.0, -1.0, 1.0))
    for shape, col in ((c_central, 0xC144EB), (c_green, 0x45E542), (c_red, 0xF55142), (c_blue, 0x4275F5),
                       (c_orange, 0xF5BF42)):
        scene.add(Object(shape).material(Material.specular(hex_color(col), 0.4)))
    scene.add(Object(plane((0.0, 1.0, 0.0), -0.80902)).material(Material.diffuse(hex_color(0xFFFFFF))))
    for x, y, z, r, e in ((-2.0, 3.5, 0.5, 0.5, 60.0), (0.0, 0.5, 5.0, 1.0, 2.0), (2.0, 1.0, -5.0, 0.6, 10.0)):
        scene.add(Light.Object(Object(sphere().scale((r, r, r)).translate((x, y, z)))
                               .material(Material.light((1.0, 1.0, 1.0), e))))
    camera = Camera.look_at((-0.9, 1.2, 2.4), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), math.pi / 4)
    return scene, camera, dict(width=1024, height=1024, max_bounces=5, num_samples=50)


# ----------------------------------------------------------------------------- the asset-driven examples
# The reference reads these files from its own examples/ directory; they are not redistributed here.  `asset(name)`
# finds them in the directories of $RPT_ASSETS (os.pathsep-separated); every scene below takes the loaded mesh as an
# argument and falls back to a seeded procedural stand-in of about the same size and extent.
def asset(name):
    """path of an asset file of the reference's examples/ (or of `name` inside pegasus.zip -> an open text file), or None"""
    import os
    for d in os.environ.get("RPT_ASSETS", "").split(os.pathsep):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


def load_asset(name):
    """the Mesh of an asset: *.obj, *.stl, or `pegasus.obj` out of pegasus.zip (examples/pegasus.rs:17-32); None if absent"""
    from . import io
    path = asset(name)
    if path is not None:
        return io.load_stl(path) if name.endswith(".stl") else io.load_obj(path)
    if name == "pegasus.obj" and asset("pegasus.zip") is not None:
        import io as _io
        import zipfile
        with zipfile.ZipFile(asset("pegasus.zip")) as z, z.open("pegasus.obj") as f:
            return io.load_obj(_io.TextIOWrapper(f, encoding="utf-8", errors="replace"))
    return None


def _cylinder_stand_in(n=42):
    """a capped cylinder with face normals in the frame of cylinder.stl (x, y in [0, 30], z in [0, 50])"""
    a = np.linspace(0.0, 2.0 * math.pi, n, endpoint=False)
    ring = np.stack([15.0 + 15.0 * np.cos(a), 15.0 + 15.0 * np.sin(a)], axis=1)
    tris = []
    for i in range(n):
        p, q = ring[i], ring[(i + 1) % n]
        b0, b1, t0, t1 = (p[0], p[1], 0.0), (q[0], q[1], 0.0), (p[0], p[1], 50.0), (q[0], q[1], 50.0)
        tris += [(b0, b1, t1), (b0, t1, t0), ((15.0, 15.0, 0.0), b1, b0), ((15.0, 15.0, 50.0), t0, t1)]
    from .shape import Triangle
    return Mesh([Triangle.from_vertices(*t) for t in tris])


def teapot(mesh=None):
    """examples/teapot.rs:8-35 (a rough red metal teapot on a plane; the Renderer defaults: 0 bounces, 1 sample)."""
    if mesh is None:
        mesh = Mesh(_teapot_stand_in(48, 8))
    scene = Scene()
    scene.add(Object(mesh.scale((0.5, 0.5, 0.5)).translate((0.0, -1.0, 0.0)))
              .material(Material.metallic_(hex_color(0xFF0000), 0.4)))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    scene.add(Light.Point((60.0, 60.0, 60.0), (0.0, 5.0, 5.0)))
    return scene, Camera(), dict(width=800, height=800, max_bounces=0, num_samples=1)


def cylinder(mesh=None):
    """examples/cylinder.rs:8-37 (an STL cylinder with the default material; point + directional light)."""
    if mesh is None:
        mesh = _cylinder_stand_in()
    scene = Scene()
    scene.add(Object(mesh.translate((-15.0, -15.0, -25.0)).scale((1.0 / 15.0, 1.0 / 15.0, 1.0 / 25.0))
                     .rotate_y(math.pi / 4.0)))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    scene.add(Light.Point((80.0, 80.0, 80.0), (0.0, 5.0, 5.0)))
    n = math.sqrt((1.0 * 1.0 + 1.0 * 1.0) + 0.0 * 0.0)
    scene.add(Light.Directional((2.0, 2.0, 2.0), (1.0 / n, -1.0 / n, 0.0 / n)))
    return scene, Camera(), dict(width=512, height=512, max_bounces=0, num_samples=1)


def rustacean(mesh=None):
    """examples/rustacean.rs:9-72 (Ferris with three glass and three metal eyes-on-stalks beads under a sphere lamp)."""
    if mesh is None:
        rows = knot_mesh(196, 32, seed=0xFE2215)   # 25 088 triangles; the asset has 25 296
        rows[:, :9] *= 1.2
        mesh = Mesh(rows).translate((0.0, 0.45, 0.0))
    cs = (2.0, 2.4, 2.0)
    scene = Scene()
    scene.add(Object(mesh.translate((0.0, 0.134649, 0.0)).scale(cs)).material(Material.specular(hex_color(0xF84C00), 0.2)))
    scene.add(Object(plane((0.0, 1.0, 0.0), 0.0)).material(Material.diffuse(hex_color(0xAAAA77))))
    balls = [(True, 0.2, (-0.81, 1.02, 0.47)), (True, 0.3, (-0.86, 1.10, 0.36)), (True, 0.4, (-0.75, 1.12, 0.34)),
             (False, 0.2, (0.87, 1.03, 0.41)), (False, 0.3, (0.75, 1.09, 0.36)), (False, 0.4, (0.85, 1.15, 0.45))]
    for is_glass, roughness, pos in balls:
        pos = (cs[0] * pos[0], cs[1] * pos[1], cs[2] * pos[2])
        scene.add(Object(sphere().scale((0.1, 0.1, 0.1)).translate(pos))
                  .material(Material.clear(1.5, roughness) if is_glass else Material.metallic_(hex_color(0xFFFFFF), roughness)))
    scene.add(Light.Object(Object(sphere().scale((2.0, 2.0, 2.0)).translate((0.0, 20.0, 3.0)))
                           .material(Material.light((1.0, 1.0, 1.0), 160.0))))
    camera = Camera.look_at((-2.5, 4.0, 8.5), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), math.pi / 6.0)
    return scene, camera, dict(width=800, height=800, max_bounces=4, num_samples=10)


def pegasus(mesh=None, hdri_size=(2048, 1024)):
    """examples/pegasus.rs:51-100 (an ice sculpture: rough transparent 100k-triangle mesh over a quad, HDRI only, EV -1.5)."""
    if mesh is None:
        rows = knot_mesh(784, 64, seed=0x9E6A)     # 100 352 triangles; pegasus.obj has 100 138
        rows[:, :9] *= 0.55
        mesh = Mesh(rows).translate((0.0, 0.72, 0.0))
    scene = Scene()
    scene.environment = Environment.Hdri(synthetic_hdri(*hdri_size))
    scene.add(Object(mesh.scale((1.4, 1.4, 1.4))).material(Material.transparent_(hex_color(0xF8F8FF), 1.31, 0.2)))
    scene.add(Object(polygon([(2.0, -0.01, 2.0), (2.0, -0.01, -2.0), (-2.0, -0.01, -2.0), (-2.0, -0.01, 2.0)]))
              .material(Material.diffuse(hex_color(0xDDDDDD))))
    camera = Camera.look_at((0.0, 1.5, 3.1), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), math.pi / 4.0)
    return scene, camera, dict(width=1200, height=1200, max_bounces=8, num_samples=10, exposure_value=-1.5)


def metal(mesh=None, hdri_size=(2048, 1024)):
    """examples/metal.rs:30-63 (two instances of ONE Arc<Mesh>: a mirror teapot and one of roughness 0.1, HDRI only)."""
    if mesh is None:
        mesh = Mesh(_teapot_stand_in(48, 8))
    scene = Scene()
    scene.environment = Environment.Hdri(synthetic_hdri(*hdri_size))
    scene.add(Object(mesh.scale((0.5, 0.5, 0.5)).translate((0.0, -1.7, 0.0))).material(Material.metallic_(hex_color(0xFFFFFF), 0.1)))
    scene.add(Object(mesh.scale((0.5, 0.5, 0.5)).translate((0.0, 0.2, 0.0))).material(Material.metallic_(hex_color(0xFFFFFF), 0.0001)))
    return scene, Camera(), dict(width=1200, height=900, max_bounces=5, num_samples=20)


def simple_video(frame=0):
    """examples/simple_video.rs:10-56, frame `frame` of 60: basic.rs's scene with the cube sliding away; the example
    rebuilds the scene and the Renderer for every frame, which is what scene_create + render_batch per frame is."""
    scene = Scene()
    scene.add(Object(sphere()))
    scene.add(Object(cube().rotate_y(math.pi / 6.0).scale((0.5, 0.3, 0.4)).translate((0.4, -0.8, 4.0 + 0.01 * float(frame))))
              .material(Material.specular(hex_color(0xFF00FF), 0.5)))
    scene.add(Object(sphere().scale((0.5, 0.5, 0.5)).translate((1.5, -0.5, 1.0))).material(Material.specular(hex_color(0x0000FF), 0.1)))
    scene.add(Object(sphere().scale((0.5, 0.5, 0.5)).translate((-1.5, -0.5, 1.0))).material(Material.specular(hex_color(0x00FF00), 0.1)))
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.specular(hex_color(0xAAAAAA), 0.5)))
    scene.add(Light.Ambient((0.01, 0.01, 0.01)))
    scene.add(Light.Point((100.0, 100.0, 100.0), (0.0, 5.0, 5.0)))
    return scene, Camera(), dict(width=800, height=600, max_bounces=1, num_samples=100)


# ----------------------------------------------------------------------------- a flat scene that is not Cornell
def polygon_room(n_walls=23, transformed_every=0, sides=(4, 5, 6, 8), seed=3):
    """Many small polygon meshes in a row (single-leaf trees), optionally with a Transformed one every few
    objects and a cube / sphere in between: the shapes the flat path kernel batches or must not batch."""
    rs = np.random.RandomState(seed)
    scene = Scene()
    for i in range(n_walls):
        k = sides[i % len(sides)]
        c = rs.uniform(-2.0, 2.0, 3)
        a = rs.randn(3)
        a /= np.linalg.norm(a)
        b = np.cross(a, rs.randn(3))
        b /= np.linalg.norm(b)
        r = rs.uniform(0.5, 1.5)
        verts = [tuple(c + r * (math.cos(t) * a + math.sin(t) * b)) for t in np.linspace(0, 2 * math.pi, k, endpoint=False)]
        shape = polygon(verts)
        if transformed_every and i % transformed_every == transformed_every - 1:
            shape = shape.rotate_y(0.3 * i).translate((0.1 * i, 0.0, -0.05 * i))
        mat = Material.diffuse(tuple(rs.uniform(0.3, 0.9, 3))) if i % 3 else Material.specular(tuple(rs.uniform(0.3, 0.9, 3)), 0.2)
        scene.add(Object(shape).material(mat))
        if i % 7 == 3:
            scene.add(Object(cube().scale((0.4, 0.4, 0.4)).translate(tuple(rs.uniform(-1.5, 1.5, 3))))
                      .material(Material.diffuse((0.8, 0.8, 0.8))))
        if i % 7 == 5:
            scene.add(Object(sphere().scale((0.3, 0.3, 0.3)).translate(tuple(rs.uniform(-1.5, 1.5, 3)))))
    quad = polygon([(-0.5, 2.9, -0.5), (-0.5, 2.9, 0.5), (0.5, 2.9, 0.5), (0.5, 2.9, -0.5)])
    scene.add(Light.Object(Object(quad).material(Material.light((1.0, 1.0, 0.9), 30.0))))
    scene.add(Light.Point((20.0, 20.0, 20.0), (0.0, 0.0, 4.0)))
    cam = Camera.look_at((0.0, 0.5, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.9)
    return scene, cam, dict(width=1920, height=1080, max_bounces=8, num_samples=512)


SCENES = {"room23": polygon_room, "sphere": sphere_scene, "cornell": cornell, "dragon": dragon,
          "fractal_spheres": fractal_spheres, "glass": glass, "wine_glass": wine_glass,
          "fractal_teapots": fractal_teapots, "basic": basic, "monomial_glass": monomial_glass, "spheres": spheres, "compound": compound,
          "teapot": teapot, "cylinder": cylinder, "rustacean": rustacean, "pegasus": pegasus, "metal": metal,
          "simple_video": simple_video}
