"""SURVEY §8f rank 3 on the GPU: geometry that comes in through the importers (reference src/io.rs:27-360) is
rendered by the HIP path and compared with the oracle fed the same triangles — bit for bit — and the C++ host
mirror (include/rpt.hpp: load_obj / load_stl / load_obj_with_mtl, examples/obj_render.cpp) produces the same frame
as the Python mirror (rpt_amd/io.py).

Assets are written here: an OBJ with `vn`, negative indices and quads; a binary and an ASCII STL (f32 coordinates);
an OBJ + MTL pair with three materials (Kd / Ns / Ni / d)."""
import math
import os
import struct
import subprocess

import numpy as np
import pytest

import rpt_amd
from rpt_amd import (Camera, GpuScene, Light, Material, Object, Scene, hex_color, io, make_params, plane, scenes, sphere)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "examples", "build")
W, H, B, SPP, SEED = 128, 72, 4, 4, 4242


def write_obj(path, rows, quads=()):
    """rows: (n, 18) triangles with normals -> `v` / `vn` / `f a//na ...`, every other face with negative indices;
    quads: lists of four points -> 4-vertex faces without normals (fan triangulation, face normals)"""
    with open(path, "w") as f:
        f.write("# written by tests/test_gpu_io.py\n")
        nv = 0
        for i, r in enumerate(rows):
            for k in range(3):
                f.write("v %r %r %r\n" % tuple(float(x) for x in r[3 * k:3 * k + 3]))
            for k in range(3):
                f.write("vn %r %r %r\n" % tuple(float(x) for x in r[9 + 3 * k:12 + 3 * k]))
            if i % 2:
                f.write("f -3//-3 -2//-2 -1//-1\n")
            else:
                f.write("f %d//%d %d//%d %d//%d\n" % (nv + 1, nv + 1, nv + 2, nv + 2, nv + 3, nv + 3))
            nv += 3
        f.write("vt 0.5 0.5\n")  # ignored with a warning by the reference (io.rs:52-55)
        for q in quads:
            for p in q:
                f.write("v %r %r %r\n" % tuple(float(x) for x in p))
            f.write("f -4 -3 -2 -1\n")


def write_stl(path, rows, ascii_):
    v = rows[:, :9].astype(np.float32)
    n = np.cross(rows[:, 3:6] - rows[:, 0:3], rows[:, 6:9] - rows[:, 0:3])
    n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    if ascii_:
        with open(path, "w") as f:
            f.write("solid knot\n")
            for a, t in zip(n, v):
                f.write("facet normal %.9g %.9g %.9g\n  outer loop\n" % tuple(a))
                for k in range(3):
                    f.write("    vertex %.9g %.9g %.9g\n" % tuple(t[3 * k:3 * k + 3]))
                f.write("  endloop\nendfacet\n")
            f.write("endsolid knot\n")
    else:
        with open(path, "wb") as f:
            f.write(b"binary stl written by tests/test_gpu_io.py".ljust(80, b" "))
            f.write(struct.pack("<I", len(v)))
            for a, t in zip(n, v):
                f.write(struct.pack("<12fH", *a, *t, 0))


def write_obj_mtl(obj_path, mtl_path, groups):
    with open(mtl_path, "w") as f:
        f.write("newmtl glassy\nKd 0.9 0.95 1.0\nNi 1.0\nNs 900\nd 0.3\n")       # Ni clamps to 1 + 1e-4, transparent
        f.write("newmtl matte\nKd 0.8 0.3 0.2\nNs 2\nillum 2\n")
        f.write("newmtl shiny\nKd 0.2 0.7 0.3\nNs 200\nNi 1.8\nd 1.0\n")
    with open(obj_path, "w") as f:
        f.write("mtllib ignored.mtl\n")
        for name, rows in groups:
            f.write("usemtl %s\n" % name)
            for r in rows:
                for k in range(3):
                    f.write("v %r %r %r\n" % tuple(float(x) for x in r[3 * k:3 * k + 3]))
                f.write("f -3 -2 -1\n")
            f.write("usemtl %s\n" % name)  # repeating the current material does not start a new object (io.rs:123)


def stage(objects):
    """the scene of examples/obj_render.cpp around the imported objects"""
    scene = Scene()
    for o in objects:
        scene.add(o)
    scene.add(Object(plane((0.0, 1.0, 0.0), -1.0)).material(Material.diffuse(hex_color(0xAAAAAA))))
    scene.add(Light.Ambient((0.02, 0.02, 0.02)))
    scene.add(Light.Object(Object(sphere().scale((1.5, 1.5, 1.5)).translate((0.0, 8.0, 3.0))).material(Material.light((1.0, 1.0, 1.0), 60.0))))
    scene.add(Light.Point((20.0, 20.0, 20.0), (-3.0, 4.0, 4.0)))
    camera = Camera.look_at((-2.5, 3.0, 5.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), math.pi / 5.0)
    return scene, camera


def assets(tmp_path):
    knot = scenes.knot_mesh(40, 8) * np.array([2.5] * 9 + [1.0] * 9)      # 640 smooth triangles
    glass = scenes.lathe_glass_mesh(12) * np.array([0.5] * 9 + [1.0] * 9)
    quads = [[(-1.5, -0.5, -1.5 + i), (-1.5, -0.5, -1.0 + i), (-1.0, 0.2, -1.0 + i), (-1.0, 0.2, -1.5 + i)] for i in range(3)]
    out = {}
    p = str(tmp_path / "knot.obj"); write_obj(p, knot, quads); out["obj"] = p
    p = str(tmp_path / "knot_bin.stl"); write_stl(p, knot, False); out["stl_binary"] = p
    p = str(tmp_path / "knot_ascii.stl"); write_stl(p, knot[:200], True); out["stl_ascii"] = p
    po, pm = str(tmp_path / "three.obj"), str(tmp_path / "three.mtl")
    third = len(knot) // 3
    write_obj_mtl(po, pm, [("glassy", glass), ("matte", knot[:third]), ("shiny", knot[third:2 * third])])
    out["obj_mtl"] = po + "+" + pm
    return out


def load_objects(kind, path):
    if kind == "obj_mtl":
        o, m = path.split("+")
        return io.load_obj_with_mtl(o, m)
    mesh = io.load_stl(path) if kind.startswith("stl") else io.load_obj(path)
    return [Object(mesh).material(Material.specular(hex_color(0xB7CA79), 0.2))]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["obj", "stl_binary", "stl_ascii", "obj_mtl"])
def test_imported_asset_renders_bit_equal_to_oracle_and_to_the_cpp_host(oracle, tmp_path, kind):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    path = assets(tmp_path)[kind]
    objects = load_objects(kind, path)
    if kind == "obj":
        assert len(objects[0].shape.triangles) == 640 + 6   # three quads -> six fan triangles
    if kind == "obj_mtl":
        assert len(objects) == 3
        assert objects[0]._material.transparent and abs(objects[0]._material.index - (1.0 + 1e-4)) < 1e-15
        assert abs(objects[1]._material.roughness - math.sqrt(math.sqrt(2.0 / 4.0))) < 1e-15
    scene, cam = stage(objects)
    p = make_params(W, H, B, SPP, seed=SEED)
    g = GpuScene(scene, 0)
    ref = oracle.OracleScene(scene).render(cam, p, threads=0)
    for flags in (0, rpt_amd._abi.RPT_FLAG_PERSISTENT, rpt_amd._abi.RPT_FLAG_WAVEFRONT):
        img = g.render_batch(cam, make_params(W, H, B, SPP, seed=SEED, flags=flags))
        assert (img == ref).all(), (kind, flags, np.abs(img - ref).max())
    # rays straight at the asset through the closest-hit entry point
    rs = np.random.RandomState(3)
    o = rs.uniform(-3, 3, (20000, 3)) + np.array([0.0, 2.0, 0.0])
    d = rs.uniform(-1.2, 1.2, (20000, 3)) - o   # aimed at the asset
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0, n0, ob0 = oracle.OracleScene(scene).closest_hit(o, d)
    t1, n1, ob1 = g.closest_hit(o, d)
    assert (t0 == t1).all() and (ob0 == ob1).all() and (n0.view(np.int64) == n1.view(np.int64)).all()
    assert (ob0 >= 0).sum() > 5000 and (ob0 == 0).sum() > 200, ((ob0 >= 0).sum(), (ob0 == 0).sum())
    g.close()
    # the same file through the C++ mirror's importer and Renderer
    prefix = str(tmp_path / ("cpp_" + kind))
    r = subprocess.run([os.path.join(BUILD, "obj_render"), path, str(W), str(H), str(B), str(SPP), str(SEED), prefix],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cpp = np.fromfile(prefix + ".f64", dtype=np.float64).reshape(-1, 3)
    assert (cpp == ref).all(), (kind, np.abs(cpp - ref).max())


def test_cpp_importers_read_what_the_python_importers_read(tmp_path):
    """no GPU needed: examples/build/io_check prints the triangles the C++ loaders produce as hex floats"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    a = assets(tmp_path)
    for kind in ("obj", "stl_binary", "stl_ascii"):
        r = subprocess.run([os.path.join(BUILD, "io_check"), a[kind]], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.split("\n")
        n = int(lines[0])
        cpp = np.array([[float.fromhex(x) for x in ln.split()] for ln in lines[1:1 + n]])
        py = load_objects(kind, a[kind])[0].shape.triangles
        py = np.asarray(py if isinstance(py, np.ndarray) else [t.row() for t in py], dtype=np.float64).reshape(-1, 18)
        assert cpp.shape == py.shape and (cpp.view(np.int64) == py.view(np.int64)).all(), kind


def test_malformed_indices_fail_like_the_reference(tmp_path):
    # io.rs:10-18: a non-positive index wraps; out of range is a panic there, an error here — never silent wrong geometry
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -5\n")
    with pytest.raises((IndexError, ValueError)):
        io.load_obj(str(p))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises((IndexError, ValueError)):
        io.load_obj(str(p))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//9\n")
    with pytest.raises((IndexError, ValueError)):
        io.load_obj(str(p))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3\n")   # a missing normal: face normal, no error
    assert len(io.load_obj(str(p)).triangles) == 1
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -5\n")
    r = subprocess.run([os.path.join(BUILD, "io_check"), str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "error" in r.stderr
