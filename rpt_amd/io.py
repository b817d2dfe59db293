"""Asset import (reference src/io.rs) — host-side, off the timed path:
`load_obj`, `load_obj_with_mtl`, `load_stl`.  Same parsing rules as the reference: faces are
fan-triangulated (io.rs:181-198), indices may be negative (io.rs:10-18), a face corner without a
normal index makes the whole triangle flat-shaded (`Triangle::from_vertices`, io.rs:186-187),
`usemtl` starts a new Object only when the material name changes (io.rs:121-135), MTL `Ns` maps to
roughness (2/(Ns+2))^(1/4), `Ni` is clamped to >= 1.0001, `d < 0.8` makes the material transparent
(io.rs:202-254); STL is binary when size == 84 + 50 n, else ASCII when it starts with "solid "
(io.rs:260-287)."""
import copy
import math
import struct

import numpy as np

from .material import Material
from .object import Object
from .shape import Mesh, Triangle


def _open(f, mode):
    return (open(f, mode), True) if isinstance(f, (str, bytes)) or hasattr(f, "__fspath__") else (f, False)


def _parse_index(value, length):  # io.rs:10-18
    try:
        index = int(value)
    except ValueError:
        return None
    return index - 1 if index > 0 else length + index


def _at(seq, idx):
    """seq[idx] as the reference indexes a Vec: a negative result of parse_index wraps to a huge usize there
    and the access panics (io.rs:14-16, 185-196); Python's negative indexing would silently pick an element
    from the end instead."""
    if not 0 <= idx < len(seq):
        raise IndexError("index out of bounds: the len is %d but the index is %d" % (len(seq), idx))
    return seq[idx]


def _point(tokens):  # parse_obj_point io.rs:150-161
    try:
        return (float(tokens[1]), float(tokens[2]), float(tokens[3]))
    except (ValueError, IndexError):
        raise ValueError("Failed to parse vertex in .OBJ")


def _face(tokens, vertices, normals):  # parse_obj_face io.rs:163-200
    vi, vni = [], []
    for vertex in tokens[1:]:
        args = (vertex.split("/") + ["", "", ""])[:3]
        v = _parse_index(args[0], len(vertices))
        if v is None:
            raise ValueError("Invalid vertex index")
        vi.append(v)
        vni.append(_parse_index(args[2], len(normals)))
    tris = []
    for i in range(1, len(vi) - 1):
        a, b, c = 0, i, i + 1
        v1, v2, v3 = _at(vertices, vi[a]), _at(vertices, vi[b]), _at(vertices, vi[c])
        if vni[a] is None or vni[b] is None or vni[c] is None:
            tris.append(Triangle.from_vertices(v1, v2, v3))
        else:
            tris.append(Triangle(v1, v2, v3, _at(normals, vni[a]), _at(normals, vni[b]), _at(normals, vni[c])))
    return tris


def _lines(f):
    for line in f:
        line = line.strip()
        if line and not line.startswith("#"):
            yield line.split()


def load_obj(file):  # io.rs:27-74
    f, own = _open(file, "r")
    try:
        vertices, normals, triangles = [], [], []
        for tokens in _lines(f):
            if tokens[0] == "v":
                vertices.append(_point(tokens))
            elif tokens[0] == "vn":
                normals.append(_point(tokens))
            elif tokens[0] == "f":
                triangles.extend(_face(tokens, vertices, normals))
        return Mesh(triangles)
    finally:
        if own:
            f.close()


def load_mtl(file):  # io.rs:202-254
    f, own = _open(file, "r")
    try:
        materials, current = {}, None
        for tokens in _lines(f):
            if tokens[0] == "newmtl":
                current = tokens[1]
                materials.setdefault(current, Material())
            else:
                if current is None:
                    raise ValueError("Material was not specified with `newmtl` before properties were added")
                mat = materials[current]
                if tokens[0] == "Kd":
                    mat.color = _point(tokens)
                elif tokens[0] == "Ns":
                    mat.roughness = math.sqrt(math.sqrt(2.0 / (float(tokens[1]) + 2.0)))
                elif tokens[0] == "Ni":
                    mat.index = max(float(tokens[1]), 1.0 + 1e-4)
                elif tokens[0] == "d":
                    if float(tokens[1]) < 0.8:
                        mat.transparent = True
        return materials
    finally:
        if own:
            f.close()


def load_obj_with_mtl(obj_file, mtl_file):  # io.rs:83-148
    materials = load_mtl(mtl_file)
    f, own = _open(obj_file, "r")
    try:
        vertices, normals, objects = [], [], []
        current_triangles, current_material, last_usemtl = [], Material(), None

        def flush():
            if current_triangles:
                objects.append(Object(Mesh(list(current_triangles))).material(copy.copy(current_material)))
                del current_triangles[:]

        for tokens in _lines(f):
            if tokens[0] == "v":
                vertices.append(_point(tokens))
            elif tokens[0] == "vn":
                normals.append(_point(tokens))
            elif tokens[0] == "f":
                current_triangles.extend(_face(tokens, vertices, normals))
            elif tokens[0] == "usemtl":
                if last_usemtl is None or last_usemtl != tokens[1]:
                    flush()
                    if tokens[1] not in materials:
                        raise ValueError("Could not found `usemtl %s` in library" % tokens[1])
                    current_material = materials[tokens[1]]
                    last_usemtl = tokens[1]
        flush()
        return objects
    finally:
        if own:
            f.close()


def load_stl(file):  # io.rs:260-360
    f, own = _open(file, "rb")
    try:
        data = f.read()
    finally:
        if own:
            f.close()
    size = len(data)
    if size < 15:
        raise ValueError("Loaded .STL file is too short")
    if size >= 84:
        (n,) = struct.unpack_from("<I", data, 80)
        if size == 84 + n * 50:  # very likely binary
            rec = np.frombuffer(data, dtype=np.dtype([("f", "<f4", (12,)), ("attr", "<u2")]), count=n, offset=84)
            v = rec["f"].astype(np.float64)  # f32 -> f64 as the reference does
            vn = v[:, 0:3]
            rows = np.concatenate([v[:, 3:6], v[:, 6:9], v[:, 9:12], vn, vn, vn], axis=1)
            return Mesh(np.ascontiguousarray(rows))
    if data[:6] == b"solid ":
        lines = data.decode("utf-8", errors="replace").split("\n")[1:]
        tris, i = [], 0
        while i < len(lines):
            line = lines[i].strip()
            if not line.startswith("facet normal "):
                if not line or line.startswith("endsolid"):
                    i += 1
                    continue
                raise ValueError("Malformed STL file: expected `facet normal`")
            vn = tuple(float(t) for t in line[len("facet normal "):].split())
            vs = []
            for j in range(3):
                vl = lines[i + 2 + j].strip()
                if not vl.startswith("vertex "):
                    raise ValueError("Malformed STL file: expected `vertex`")
                vs.append(tuple(float(t) for t in vl[len("vertex "):].split()))
            tris.append(Triangle(vs[0], vs[1], vs[2], vn, vn, vn))
            i += 7
        return Mesh(tris)
    raise ValueError("Loaded .STL file, but could not determine format")
