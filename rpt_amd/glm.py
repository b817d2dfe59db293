"""The handful of nalgebra-glm 0.10 functions the reference's scene construction uses
(reference src/shape.rs:111-124, 202-284), in plain Python floats (IEEE f64, no FMA).

Matrices are flat column-major lists, as nalgebra stores them: element (r, c) of a 4x4 is
m[c*4+r].  These run on the host at scene-build time only (never in the timed path); their
results travel through the C ABI inside `RptTransform`, so every consumer of the ABI reads
identical bits whatever rounding happens here.
"""
import math


def identity4():
    return [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]


def mul4(a, b):
    """a * b, accumulated column by column the way nalgebra's gemv does."""
    out = [0.0] * 16
    for j in range(4):
        for r in range(4):
            acc = a[0 * 4 + r] * b[j * 4 + 0]
            acc = acc + a[1 * 4 + r] * b[j * 4 + 1]
            acc = acc + a[2 * 4 + r] * b[j * 4 + 2]
            acc = acc + a[3 * 4 + r] * b[j * 4 + 3]
            out[j * 4 + r] = acc
    return out


def translation(v):
    m = identity4()
    m[12], m[13], m[14] = float(v[0]), float(v[1]), float(v[2])
    return m


def scaling(v):
    m = identity4()
    m[0], m[5], m[10] = float(v[0]), float(v[1]), float(v[2])
    return m


def rotation(angle, axis):
    """glm::rotate(identity, angle, axis): Rotation3::from_axis_angle(normalize(axis), angle)."""
    n = math.sqrt((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2])
    ux, uy, uz = axis[0] / n, axis[1] / n, axis[2] / n
    sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
    s, c = math.sin(angle), math.cos(angle)
    omc = 1.0 - c
    rows = [
        [sqx + (1.0 - sqx) * c, ux * uy * omc - uz * s, ux * uz * omc + uy * s],
        [ux * uy * omc + uz * s, sqy + (1.0 - sqy) * c, uy * uz * omc - ux * s],
        [ux * uz * omc - uy * s, uy * uz * omc + ux * s, sqz + (1.0 - sqz) * c],
    ]
    m = identity4()
    for r in range(3):
        for col in range(3):
            m[col * 4 + r] = rows[r][col]
    return m


def mat4_to_mat3(m):
    return [m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]]


def determinant3(m):
    m11, m21, m31, m12, m22, m32, m13, m23, m33 = m
    minor_m12_m23 = m22 * m33 - m32 * m23
    minor_m11_m23 = m21 * m33 - m31 * m23
    minor_m11_m22 = m21 * m32 - m31 * m22
    return m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22


def inverse3(m):
    """nalgebra try_inverse for 3x3 (closed form); zeros if singular."""
    m11, m21, m31, m12, m22, m32, m13, m23, m33 = m
    minor_m12_m23 = m22 * m33 - m32 * m23
    minor_m11_m23 = m21 * m33 - m31 * m23
    minor_m11_m22 = m21 * m32 - m31 * m22
    det = m11 * minor_m12_m23 - m12 * minor_m11_m23 + m13 * minor_m11_m22
    if det == 0.0:
        return [0.0] * 9
    r = [[0.0] * 3 for _ in range(3)]
    r[0][0] = minor_m12_m23 / det
    r[0][1] = (m13 * m32 - m33 * m12) / det
    r[0][2] = (m12 * m23 - m22 * m13) / det
    r[1][0] = -minor_m11_m23 / det
    r[1][1] = (m11 * m33 - m31 * m13) / det
    r[1][2] = (m13 * m21 - m23 * m11) / det
    r[2][0] = minor_m11_m22 / det
    r[2][1] = (m12 * m31 - m32 * m11) / det
    r[2][2] = (m11 * m22 - m21 * m12) / det
    return [r[row][col] for col in range(3) for row in range(3)]


def inverse_transpose3(m):
    inv = inverse3(m)
    return [inv[r * 3 + c] for c in range(3) for r in range(3)]  # transpose


def inverse4(m):
    """nalgebra do_inverse4 (the MESA gluInvertMatrix cofactor expansion)."""
    o = [0.0] * 16
    o[0] = (m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15]
            + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10])
    o[4] = (-m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15]
            - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10])
    o[8] = (m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15]
            + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9])
    o[12] = (-m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14]
             - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9])
    o[1] = (-m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15]
            - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10])
    o[5] = (m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15]
            + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10])
    o[9] = (-m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15]
            - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9])
    o[13] = (m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14]
             + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9])
    o[2] = (m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15]
            + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6])
    o[6] = (-m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15]
            - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6])
    o[10] = (m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15]
             + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5])
    o[14] = (-m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14]
             - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5])
    o[3] = (-m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11]
            - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6])
    o[7] = (m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11]
            + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6])
    o[11] = (-m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11]
             - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5])
    o[15] = (m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10]
             + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5])
    det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12]
    if det == 0.0:
        return [0.0] * 16
    inv_det = 1.0 / det
    return [x * inv_det for x in o]


def vec3(x, y, z):
    return (float(x), float(y), float(z))


def dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def scale(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _ieee_div(a, b):
    """a / b as IEEE (and Rust) define it: Python raises on a zero divisor instead."""
    if b != 0.0:
        return a / b
    if a == 0.0 or a != a:
        return math.nan
    return math.copysign(math.inf, a) * math.copysign(1.0, b)


def normalize(a):
    # glm::normalize of a zero vector is (NaN, NaN, NaN) in the reference (a degenerate triangle's normal,
    # mesh.rs:25-36), not an error
    n = math.sqrt(dot(a, a))
    return (_ieee_div(a[0], n), _ieee_div(a[1], n), _ieee_div(a[2], n))
