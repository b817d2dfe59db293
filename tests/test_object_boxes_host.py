"""The object filter's host tables (rpt_amd/csrc/host_scene.cpp fill_object_boxes), checked without a GPU: the PRODUCT's
flattener, compiled with g++ next to a small C++ driver (tests/cpp/object_boxes_host.cpp), flattens a scene that holds
one object of every kind the filter must leave alone — an unbounded Plane, a mesh with a sliver triangle, a placement with
condition number 1e5, a sphere smaller than 64 grid steps — among ordinary spheres, cubes and meshes; the exemption mask,
and that every filtered object's decoded 16-bit box contains its bounding box (Transformed::bounding_box,
shape.rs:153-176) with the margin of a grid step the device's f32 arithmetic relies on."""
import math
import os
import subprocess

import numpy as np

import rpt_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _world_box(shape, half):
    """the box of the eight transformed corners of [-half, half]^3 (column-major 4x4, shape.rs:153-176)"""
    m = np.array(shape.transform_m).reshape(4, 4).T
    corners = np.array([[sx, sy, sz, 1.0] for sx in (-half, half) for sy in (-half, half) for sz in (-half, half)])
    w = (corners @ m.T)[:, :3]
    return w.min(0), w.max(0)


def test_object_filter_tables_exemptions_and_margins(tmp_path):
    exe = str(tmp_path / "object_boxes_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "object_boxes_host.cpp"),
                           os.path.join(ROOT, "rpt_amd", "csrc", "host_scene.cpp"), "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, check=True).stdout.splitlines()
    head = out[0].split()
    assert head[:2] == ["rc", "0"] and head[3] == "1" and head[5] == "9", out[0]
    always = int(head[7], 16)
    # plane (0), sliver mesh (4), ill-conditioned placement (5), tiny sphere (6) are never filtered; the others are
    assert always == (1 << 0) | (1 << 4) | (1 << 5) | (1 << 6), hex(always)
    grid = np.array([float(x) for x in out[1].split()[1:]])
    qlo, step = grid[0:3], grid[3:6]
    boxes = {}
    for line in out[2:11]:
        f = line.split()
        boxes[int(f[1])] = (int(f[3]), np.array([int(x) for x in f[5:11]], dtype=np.float64))
    for i in (0, 4, 5, 6):
        full, q = boxes[i]
        assert full == 1 and (q[:3] <= 0).all() and (q[3:] >= 65535).all()  # centre -+ half-extent covers the whole grid
    # the filtered ones: decoded box ⊇ bounding box + (almost) one step on every side
    sph1 = rpt_amd.sphere().scale((0.5, 0.5, 0.5)).translate((1.0, 0.0, 0.0))
    cub2 = rpt_amd.cube().rotate_y(0.7).scale((0.5, 1.5, 0.5)).translate((2.5, 0.0, -2.5))
    sph7 = rpt_amd.sphere().translate((3.0, 2.0, 3.0))
    tri8 = np.array([(-3.0, 0.0, -3.0), (-3.0, 2.0, -3.0), (-3.0, 2.0, 3.0)])
    rot = np.array(rpt_amd.polygon([tuple(v) for v in tri8]).rotate_y(0.3).transform_m).reshape(4, 4).T
    tlo, thi = tri8.min(0), tri8.max(0)   # Transformed<Mesh>: the mesh's box, then the eight corners (not the vertices)
    c8 = np.array([[x, y, z, 1.0] for x in (tlo[0], thi[0]) for y in (tlo[1], thi[1]) for z in (tlo[2], thi[2])]) @ rot.T
    expect = {1: _world_box(sph1, 1.0), 2: _world_box(cub2, 0.5), 7: _world_box(sph7, 1.0),
              3: (np.array([-2.0, 0.0, -2.0]), np.array([2.0, 0.0, 2.0])), 8: (c8[:, :3].min(0), c8[:, :3].max(0))}
    for i, (lo, hi) in expect.items():
        full, q = boxes[i]
        assert full == 0
        dlo, dhi = qlo + q[:3] * step, qlo + q[3:] * step
        assert (dlo <= lo - 0.999 * step).all() and (dhi >= hi + 0.999 * step).all(), (i, dlo, lo, dhi, hi)
        assert (dlo >= lo - 3.0 * step).all() and (dhi <= hi + 3.0 * step).all(), i   # and not needlessly large
    # the grid spans the bounded, filtered objects with two steps of padding (coordinates map to [2, 65531])
    all_lo = np.min([expect[i][0] for i in expect], axis=0)
    all_hi = np.max([expect[i][1] for i in expect], axis=0)
    assert np.allclose(qlo, all_lo - 2.0 * step, rtol=0, atol=1e-9) and np.allclose(step, (all_hi - all_lo) / 65529.0, rtol=1e-12)
    # smallest semi-axis of the tiny sphere against 64 steps (quadric_too_small), as the header of the C++ file claims
    assert 2e-3 / math.sqrt(3.0) < 64.0 * step.max() < 0.5 / math.sqrt(3.0)
    # a group among the top-level objects: no filter for that scene
    assert out[11].split() == ["with_group", "rc", "0", "ok", "0"]
    # group children: the tiny sphere (child 20) and the monomial surface (child 21) are never filtered, the others are
    assert out[12].split()[:3] == ["group", "rc", "0"]
    refs = [(int(f[2]), int(f[4])) for f in (line.split() for line in out[13:]) if f and f[0] == "ref"]
    assert len(refs) >= 22 and {c for c, _ in refs} == set(range(22))
    for child, full in refs:
        assert full == (1 if child in (20, 21) else 0), (child, full)
