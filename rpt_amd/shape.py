"""Host-side mirror of the reference's shape layer (src/shape.rs, src/shape/*.rs,
src/kdtree.rs) — DESCRIPTION ONLY.  No intersection code lives here: these classes record
what the user built and lower it to the C ABI's `RptShape` (include/rpt_gpu.h); all geometry
work happens in the HIP kernels behind `librptgpu.so`.

The reference's `dyn Shape` is an open trait (shape.rs:18-25); the device understands the
closed set below.  Anything else raises at lowering time.
"""
import ctypes as C

import numpy as np

from . import _abi, glm


class Shape:
    """Base of the closed shape set; carries the `Transformable` builder API
    (shape.rs:179-230): every method returns a `Transformed` wrapping `self`."""

    def translate(self, v):
        return Transformed(self, glm.translation(v))

    def scale(self, v):
        return Transformed(self, glm.scaling(v))

    def rotate(self, angle, axis):
        return Transformed(self, glm.rotation(angle, axis))

    def rotate_x(self, angle):
        return Transformed(self, glm.rotation(angle, (1.0, 0.0, 0.0)))

    def rotate_y(self, angle):
        return Transformed(self, glm.rotation(angle, (0.0, 1.0, 0.0)))

    def rotate_z(self, angle):
        return Transformed(self, glm.rotation(angle, (0.0, 0.0, 1.0)))

    def transform(self, m):
        return Transformed(self, [float(x) for x in m])

    # -- lowering ---------------------------------------------------------------------
    def _fill(self, out, keep):
        raise NotImplementedError

    def lower(self, keep):
        """-> RptShape; `keep` collects every ctypes buffer that must outlive the call."""
        s = _abi.RptShape()
        self._fill(s, keep)
        return s


class Sphere(Shape):  # src/shape/sphere.rs:9
    def _fill(self, s, keep):
        s.kind = _abi.RPT_SHAPE_SPHERE


class Cube(Shape):  # src/shape/cube.rs:8
    def _fill(self, s, keep):
        s.kind = _abi.RPT_SHAPE_CUBE


class Plane(Shape):  # src/shape/plane.rs:7-13
    def __init__(self, normal, value):
        self.normal = glm.vec3(*normal)
        self.value = float(value)

    def _fill(self, s, keep):
        s.kind = _abi.RPT_SHAPE_PLANE
        s.plane_normal[:] = self.normal
        s.plane_value = self.value


class MonomialSurface(Shape):  # src/shape/monomial_surface.rs:12-18
    """y = height * sqrt(x^2 + z^2)^exp over the unit disc; like the reference (monomial_surface.rs:10)
    intersection and normals are only valid for exp = 4."""

    def __init__(self, height, exp):
        self.height = float(height)
        self.exp = float(exp)

    def _fill(self, s, keep):
        s.kind = _abi.RPT_SHAPE_MONOMIAL
        s.monomial_height = self.height
        s.monomial_exp = self.exp

    def closest_point(self, point, steps=100):
        """monomial_surface.rs:126-152 (steps=100) / :154-181 `closest_point_precise` (steps=10000)."""
        import math
        x, y, z = (float(c) for c in point)
        if math.sqrt((x * x + y * y) + z * z) < 1e-12:
            return (x, y, z)
        px, py = math.hypot(x, z), y
        best, best_x = 1e18, -1.0
        for i in range(-steps, steps + 1):
            xf = i / float(steps)
            x4 = (xf * xf) * (xf * xf)
            dx, dy = px - xf, py - self.height * x4
            d2 = dx * dx + dy * dy
            if d2 < best:
                best, best_x = d2, xf
        n = math.sqrt(x * x + z * z)

        def div(a, b):  # IEEE division: a point on the axis normalises to NaN, as in the reference
            return a / b if b != 0.0 else (math.nan if a == 0.0 or a != a else math.copysign(math.inf, a))

        qx, qz = best_x * div(x, n), best_x * div(z, n)
        r2 = qx * qx + qz * qz
        return (qx, self.height * (r2 * r2), qz)

    def closest_point_precise(self, point):
        return self.closest_point(point, steps=10000)


class Triangle(Shape):  # src/shape/mesh.rs:8-22
    def __init__(self, v1, v2, v3, n1, n2, n3):
        self.v1, self.v2, self.v3 = glm.vec3(*v1), glm.vec3(*v2), glm.vec3(*v3)
        self.n1, self.n2, self.n3 = glm.vec3(*n1), glm.vec3(*n2), glm.vec3(*n3)

    @staticmethod
    def from_vertices(v1, v2, v3):  # mesh.rs:26-36
        v1, v2, v3 = glm.vec3(*v1), glm.vec3(*v2), glm.vec3(*v3)
        n = glm.normalize(glm.cross(glm.sub(v2, v1), glm.sub(v3, v1)))
        return Triangle(v1, v2, v3, n, n, n)

    def row(self):
        return self.v1 + self.v2 + self.v3 + self.n1 + self.n2 + self.n3

    def _fill(self, s, keep):
        raise _abi.RptGpuError(_abi.RPTGPU_E_UNSUPPORTED_SHAPE,
                               "a bare Triangle is only supported inside a Mesh")


class KdTree(Shape):
    """`KdTree<T>` (kdtree.rs:100-119).  `KdTree<Triangle>` is `Mesh` (mesh.rs:102); a tree
    of other bounded shapes is `KdTree<Box<dyn Bounded>>` (examples/fractal_spheres.rs:45).
    The tree itself is built inside the library by the reference rule (kdtree.rs:235-345)."""

    def __init__(self, objects):
        self.triangles = None
        self.objects = None
        if isinstance(objects, np.ndarray):  # fast path: (n, 18) array of triangle rows
            arr = np.ascontiguousarray(objects, dtype=np.float64)
            assert arr.ndim == 2 and arr.shape[1] == 18
            self.triangles = arr
        else:
            objects = list(objects)
            if objects and all(isinstance(o, Triangle) for o in objects):
                self.triangles = np.array([o.row() for o in objects], dtype=np.float64)
            elif not objects:
                self.triangles = np.zeros((0, 18), dtype=np.float64)
            else:
                self.objects = objects

    def __len__(self):
        return len(self.triangles) if self.triangles is not None else len(self.objects)

    def _fill(self, s, keep):
        if self.triangles is not None:
            s.kind = _abi.RPT_SHAPE_MESH
            keep.append(self.triangles)
            s.triangles = self.triangles.ctypes.data_as(C.POINTER(_abi.RptTriangle))
            s.num_triangles = len(self.triangles)
        else:
            s.kind = _abi.RPT_SHAPE_GROUP
            arr = (_abi.RptShape * len(self.objects))()
            for i, o in enumerate(self.objects):
                if isinstance(o, Plane) or (isinstance(o, Transformed) and isinstance(o.shape, Plane)):
                    raise _abi.RptGpuError(_abi.RPTGPU_E_UNSUPPORTED_SHAPE,
                                           "Plane is not Bounded (kdtree.rs:9-12)")
                o._fill(arr[i], keep)
            keep.append(arr)
            s.children = C.cast(arr, C.POINTER(_abi.RptShape))
            s.num_children = len(self.objects)


def Mesh(triangles):  # `pub type Mesh = KdTree<Triangle>` (mesh.rs:102); Mesh::new
    return KdTree(triangles)


class Transformed(Shape):
    """`Transformed<T>` (shape.rs:101-124): precomputes the same five fields.  Chained
    transforms compose as T_new * M_old without nesting (shape.rs:234-284)."""

    def __init__(self, shape, transform):
        self.shape = shape
        self.transform_m = list(transform)
        self.inverse_transform = glm.inverse4(self.transform_m)
        self.linear = glm.mat4_to_mat3(self.transform_m)
        self.scale_det = glm.determinant3(self.linear)
        self.normal_transform = glm.inverse_transpose3(self.linear)

    def translate(self, v):
        return Transformed(self.shape, glm.mul4(glm.translation(v), self.transform_m))

    def scale(self, v):
        return Transformed(self.shape, glm.mul4(glm.scaling(v), self.transform_m))

    def rotate(self, angle, axis):
        return Transformed(self.shape, glm.mul4(glm.rotation(angle, axis), self.transform_m))

    def rotate_x(self, angle):
        return self.rotate(angle, (1.0, 0.0, 0.0))

    def rotate_y(self, angle):
        return self.rotate(angle, (0.0, 1.0, 0.0))

    def rotate_z(self, angle):
        return self.rotate(angle, (0.0, 0.0, 1.0))

    def transform(self, m):
        return Transformed(self.shape, glm.mul4([float(x) for x in m], self.transform_m))

    def _fill(self, s, keep):
        if isinstance(self.shape, Transformed):
            raise _abi.RptGpuError(_abi.RPTGPU_E_UNSUPPORTED_SHAPE, "nested Transformed")
        self.shape._fill(s, keep)
        s.transformed = 1
        s.xf.transform[:] = self.transform_m
        s.xf.linear[:] = self.linear
        s.xf.inverse_transform[:] = self.inverse_transform
        s.xf.normal_transform[:] = self.normal_transform
        s.xf.scale = self.scale_det


# helper constructors, shape.rs:286-313
def sphere():
    return Sphere()


def monomial_surface(height, exp):  # shape.rs:292-294
    return MonomialSurface(height, exp)


def plane(normal, value):
    return Plane(normal, value)


def cube():
    return Cube()


def polygon(verts):  # shape.rs:307-313 (triangle fan)
    tris = [Triangle.from_vertices(verts[0], verts[i], verts[i + 1]) for i in range(1, len(verts) - 1)]
    return Mesh(tris)
