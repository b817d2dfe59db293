"""Asset import mirrors reference src/io.rs: fan triangulation, negative indices, per-corner
normals vs flat fallback, usemtl grouping, MTL conversions, binary / ASCII STL detection."""
import struct

import numpy as np
import pytest

import rpt_amd
from rpt_amd import load_obj, load_obj_with_mtl, load_stl, scenes

OBJ = """# a quad (fan -> 2 triangles), a triangle with normals, a triangle with negative indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vn 0 0 1
vt 0.5 0.5
f 1 2 3 4
f 1//1 2//1 3//1
v 2 0 0
v 3 0 0
v 3 1 0
f -3 -2 -1
f 1/1/1 2/1 3/1/1
"""


def test_load_obj_rules(tmp_path):
    p = tmp_path / "a.obj"
    p.write_text(OBJ)
    m = load_obj(str(p))
    t = m.triangles
    assert t.shape == (5, 18)
    # quad fan: (v1,v2,v3), (v1,v3,v4)  (io.rs:181-198)
    assert (t[0, :9] == [0, 0, 0, 1, 0, 0, 1, 1, 0]).all() and (t[1, :9] == [0, 0, 0, 1, 1, 0, 0, 1, 0]).all()
    assert (t[0, 9:] == [0, 0, 1] * 3).all()  # no normal indices -> Triangle::from_vertices
    assert (t[2, 9:] == [0, 0, 1] * 3).all()  # explicit normals
    assert (t[3, :9] == [2, 0, 0, 3, 0, 0, 3, 1, 0]).all()  # negative indices count from the end (io.rs:10-18)
    assert (t[4, 9:] == [0, 0, 1] * 3).all()  # one corner without a normal -> the whole triangle is flat
    with open(p) as f:
        assert (load_obj(f).triangles == t).all()  # file objects work too


def test_obj_roundtrip_of_a_generated_mesh(tmp_path):
    rows = scenes.knot_mesh(16, 6)
    lines = []
    for r in rows:
        for k in range(3):
            lines.append("v %r %r %r" % tuple(float(x) for x in r[3 * k:3 * k + 3]))
        for k in range(3):
            lines.append("vn %r %r %r" % tuple(float(x) for x in r[9 + 3 * k:12 + 3 * k]))
        lines.append("f -3//-3 -2//-2 -1//-1")
    p = tmp_path / "knot.obj"
    p.write_text("\n".join(lines) + "\n")
    assert (load_obj(str(p)).triangles == rows).all()  # repr() round-trips doubles exactly


def test_load_obj_with_mtl(tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl red\nKd 0.8 0.1 0.1\nNs 30\nNi 1.0\n\nnewmtl glass\nKd 1 1 1\nd 0.5\nNi 1.5\nKa 1 1 1\n")
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nusemtl red\nf 1 2 3\nusemtl red\nf 1 2 4\nusemtl glass\nf 2 3 4\n")
    objs = load_obj_with_mtl(str(tmp_path / "m.obj"), str(tmp_path / "m.mtl"))
    assert len(objs) == 2 and len(objs[0].shape) == 2 and len(objs[1].shape) == 1  # repeated usemtl does not split
    red, glass = objs[0]._material, objs[1]._material
    assert red.color == (0.8, 0.1, 0.1) and abs(red.roughness - (2.0 / 32.0) ** 0.25) < 1e-15
    assert red.index == 1.0 + 1e-4 and not red.transparent  # Ni clamped (io.rs:238-240)
    assert glass.transparent and glass.index == 1.5
    with pytest.raises(ValueError):
        (tmp_path / "bad.obj").write_text("v 0 0 0\nusemtl nope\n")
        load_obj_with_mtl(str(tmp_path / "bad.obj"), str(tmp_path / "m.mtl"))
    # the loaded objects go straight into a scene description
    scene = rpt_amd.Scene()
    for o in objs:
        scene.add(o)
    desc, keep = scene.lower()
    assert desc.num_objects == 2 and desc.objects[1].material.transparent == 1


def test_load_stl_binary_and_ascii(tmp_path):
    tris = [((0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0)), ((1, 0, 0), (0.5, 0.25, 2), (0, 1, 1), (0, 0, 1))]
    b = bytearray(b"solid looks-like-ascii-but-is-binary".ljust(80, b" ")) + struct.pack("<I", len(tris))
    for n, a, c, d in tris:
        b += struct.pack("<12fH", *n, *a, *c, *d, 0)
    p = tmp_path / "b.stl"
    p.write_bytes(bytes(b))
    m = load_stl(str(p))  # size rule wins over the "solid" header (io.rs:265-274)
    assert m.triangles.shape == (2, 18)
    assert (m.triangles[1, :9] == [0.5, 0.25, 2, 0, 1, 1, 0, 0, 1]).all() and (m.triangles[1, 9:] == [1, 0, 0] * 3).all()
    a = "solid t\n" + "".join(
        " facet normal %g %g %g\n  outer loop\n   vertex %g %g %g\n   vertex %g %g %g\n   vertex %g %g %g\n  endloop\n endfacet\n"
        % (n + v1 + v2 + v3) for n, v1, v2, v3 in tris) + "endsolid t\n"
    pa = tmp_path / "a.stl"
    pa.write_text(a)
    ma = load_stl(str(pa))
    assert (ma.triangles == m.triangles).all()
    with pytest.raises(ValueError):
        (tmp_path / "c.stl").write_bytes(b"garbage that is long enough")
        load_stl(str(tmp_path / "c.stl"))


# ---- the reference's own asset files, when the reference tree is present (this container only;
# ---- the GPU box has no /root/reference, and these tests need no GPU)
REF_EXAMPLES = "/root/reference/examples"


@pytest.mark.skipif(not __import__("os").path.isdir(REF_EXAMPLES), reason="reference tree not present")
@pytest.mark.parametrize("name,kind", [("teapot.obj", "obj"), ("wine_glass.obj", "obj"), ("rustacean.obj", "obj"),
                                       ("monomial.obj", "obj"), ("cylinder.stl", "stl")])
def test_reference_assets_load_and_build(name, kind):
    import os

    from oracle import oracle_ffi as O
    path = os.path.join(REF_EXAMPLES, name)
    mesh = load_obj(path) if kind == "obj" else load_stl(path)
    t = mesh.triangles
    assert t.ndim == 2 and t.shape[1] == 18 and len(t) > 50
    assert np.isfinite(t).all()
    nrm = np.linalg.norm(t[:, 9:].reshape(-1, 3), axis=1)
    assert (np.abs(nrm - 1.0) < 1e-3).mean() > 0.99  # 6-decimal file normals (passed through, io.rs:49-53) or from_vertices
    # the face count of the file is the triangle count after fan triangulation (io.rs:181-198)
    if kind == "obj":
        fans = 0
        with open(path) as f:
            for line in f:
                tok = line.split()
                if tok and tok[0] == "f":
                    fans += len(tok) - 3
        assert fans == len(t)
    # the product's kd builder and the restatement of KdTree::new agree on the real asset
    lo = np.minimum(np.minimum(t[:, 0:3], t[:, 3:6]), t[:, 6:9])
    hi = np.maximum(np.maximum(t[:, 0:3], t[:, 3:6]), t[:, 6:9])
    boxes = np.concatenate([lo, hi], axis=1)
    from rpt_amd import _abi
    from rpt_amd.device import kdtree_build
    a = kdtree_build(boxes, _abi.load_library(), "rptgpu")
    b = kdtree_build(boxes, O.lib(), "oracle")
    assert a["max_depth"] == b["max_depth"] and a["max_depth"] <= 32
    for k in ("split", "info", "a", "b", "refs"):
        assert a[k].shape == b[k].shape and (a[k] == b[k]).all(), k


def _cpp_io_check():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "build", "io_check")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples"), "build/io_check"])
    return exe


def _cpp_rows(path):
    import subprocess
    out = subprocess.check_output([_cpp_io_check(), str(path)], text=True).split("\n")
    n = int(out[0])
    rows = np.array([[float.fromhex(x) for x in line.split()] for line in out[1:1 + n]], dtype=np.float64).reshape(n, 18)
    return rows


def test_cpp_mirror_loaders_equal_the_python_mirror(tmp_path):
    # include/rpt.hpp load_obj / load_stl (reference src/io.rs) against rpt_amd.io, bit for bit
    p = tmp_path / "a.obj"
    p.write_text(OBJ)
    assert (_cpp_rows(p) == load_obj(str(p)).triangles).all()
    rows = scenes.knot_mesh(12, 5)
    q = tmp_path / "b.stl"
    with open(q, "wb") as f:
        f.write(b"\0" * 80 + struct.pack("<I", len(rows)))
        for r in rows:
            f.write(struct.pack("<12fH", *r[9:12].astype(np.float32), *r[0:9].astype(np.float32), 0))
    assert (_cpp_rows(q) == load_stl(str(q)).triangles).all()
    a = tmp_path / "c.stl"
    a.write_text("solid t\n" + "".join(
        "facet normal %r %r %r\n outer loop\n  vertex %r %r %r\n  vertex %r %r %r\n  vertex %r %r %r\n endloop\nendfacet\n"
        % tuple(float(x) for x in np.concatenate([r[9:12], r[0:9]])) for r in rows[:7]) + "endsolid t\n")
    assert (_cpp_rows(a) == load_stl(str(a)).triangles).all()
    import os
    for name in ("teapot.obj", "cylinder.stl"):  # the reference's own assets, when present
        path = os.path.join(REF_EXAMPLES, name)
        if os.path.exists(path):
            ref = (load_obj(path) if name.endswith(".obj") else load_stl(path)).triangles
            assert (_cpp_rows(path) == ref).all(), name


@pytest.mark.skipif(not __import__("os").path.isdir(REF_EXAMPLES), reason="reference tree not present")
@pytest.mark.parametrize("name,asset", [("teapot", "teapot.obj"), ("cylinder", "cylinder.stl"), ("rustacean", "rustacean.obj"),
                                        ("metal", "teapot.obj"), ("wine_glass", "wine_glass.obj"), ("fractal_teapots", "teapot.obj"),
                                        ("pegasus", "pegasus.obj")])
def test_example_scenes_accept_the_reference_assets(name, asset, monkeypatch):
    """rpt_amd.scenes' transcriptions of the asset-driven examples, fed the reference's OWN files through
    scenes.load_asset ($RPT_ASSETS; pegasus.obj comes out of pegasus.zip as in examples/pegasus.rs:17-32): the scene
    passes the product's flattener (scene_create answers NO_DEVICE here, i.e. validation went through) and the oracle
    renders a small frame of it with finite values.  (GPU vs oracle on these files: scripts/real_assets.py.)"""
    import rpt_amd
    from oracle import oracle_ffi as O
    from rpt_amd import GpuScene, _abi, make_params, scenes
    monkeypatch.setenv("RPT_ASSETS", REF_EXAMPLES)
    mesh = scenes.load_asset(asset)
    assert mesh is not None and len(mesh.triangles) > 300
    kw = dict(hdri_size=(64, 32)) if name in ("metal", "wine_glass", "pegasus") else {}
    scene, cam, cfg = scenes.SCENES[name](mesh=mesh, **kw)
    try:
        GpuScene(scene, 0).close()
    except rpt_amd.RptGpuError as e:
        assert e.code == _abi.RPTGPU_E_NO_DEVICE, e
    p = make_params(24, 24, min(cfg["max_bounces"], 3), 2, seed=5, exposure_value=cfg.get("exposure_value", 0.0))
    img = O.OracleScene(scene).render(cam, p, threads=0)
    assert img.shape == (24 * 24, 3) and np.isfinite(img).all() and img.max() > 0.0
    monkeypatch.delenv("RPT_ASSETS")
    assert scenes.load_asset(asset) is None   # without $RPT_ASSETS the scenes use their stand-ins
