"""Build-time math (float64 numpy) used while assembling a model: quaternions are (x, y, z, w),
transforms are (position, quaternion) pairs -- the conventions of dflex/dflex/util.py:47-239, which
must be matched exactly so that the model constants come out identical to the reference's."""
import math

import numpy as np


def normalize(v):
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v)
    return v if n == 0.0 else v / n


def quat_identity():
    return np.array([0.0, 0.0, 0.0, 1.0])


def quat_from_axis_angle(axis, angle):
    a = np.array(axis, dtype=np.float64)
    h = 0.5 * angle
    s = math.sin(h)
    return np.array([a[0] * s, a[1] * s, a[2] * s, math.cos(h)])


def quat_inverse(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def quat_multiply(a, b):
    return np.array([a[3] * b[0] + b[3] * a[0] + a[1] * b[2] - b[1] * a[2],
                     a[3] * b[1] + b[3] * a[1] + a[2] * b[0] - b[2] * a[0],
                     a[3] * b[2] + b[3] * a[2] + a[0] * b[1] - b[0] * a[1],
                     a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def quat_rotate(q, x):
    x = np.asarray(x, dtype=np.float64)
    v = np.array([q[0], q[1], q[2]])
    return x * (2.0 * q[3] * q[3] - 1.0) + np.cross(v, x) * q[3] * 2.0 + v * np.dot(v, x) * 2.0


def quat_to_matrix(q):
    cols = [quat_rotate(q, e) for e in np.eye(3)]
    return np.array(cols).T


def rpy2quat(roll, pitch, yaw):
    cy, sy = math.cos(yaw * 0.5), math.sin(yaw * 0.5)
    cr, sr = math.cos(roll * 0.5), math.sin(roll * 0.5)
    cp, sp = math.cos(pitch * 0.5), math.sin(pitch * 0.5)
    return (cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
            cy * cr * cp + sy * sr * sp)


def transform(p, q):
    return (np.array(p, dtype=np.float64), np.array(q, dtype=np.float64))


def transform_identity():
    return (np.zeros(3), quat_identity())


def transform_point(t, p):
    return np.asarray(t[0], dtype=np.float64) + quat_rotate(t[1], p)


def transform_multiply(t, u):
    return (quat_rotate(t[1], u[0]) + t[0], quat_multiply(t[1], u[1]))


def transform_flatten(t):
    return np.array([*t[0], *t[1]], dtype=np.float64)


def shifted_inertia(m, I, p, q):
    """Inertia `I` of mass `m`, rotated by `q` and shifted by `p` (Steiner).

    NOTE: `R * I * R.T` is an ELEMENTWISE product, exactly as in the reference
    (dflex/dflex/util.py:235-239 `transform_inertia`); a matrix product would be the textbook formula,
    but the shipped models (and therefore every published result) were built with this one, so the
    constants are reproduced as they are."""
    R = quat_to_matrix(q)
    p = np.asarray(p, dtype=np.float64)
    return R * I * R.T + m * (np.dot(p, p) * np.eye(3) - np.outer(p, p))


def transform_inverse(t):
    qi = quat_inverse(t[1])
    return (-quat_rotate(qi, t[0]), qi)


def quat_from_matrix(m):
    """Rotation matrix -> unit quaternion (Shepperd's method, largest-pivot branch selection)."""
    m = np.asarray(m, dtype=np.float64)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr >= 0.0:
        h = math.sqrt(tr + 1.0)
        w = 0.5 * h
        h = 0.5 / h
        x, y, z = (m[2, 1] - m[1, 2]) * h, (m[0, 2] - m[2, 0]) * h, (m[1, 0] - m[0, 1]) * h
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        h = math.sqrt((m[i, i] - (m[j, j] + m[k, k])) + 1.0)
        v = [0.0, 0.0, 0.0]
        v[i] = 0.5 * h
        h = 0.5 / h
        v[j] = (m[i, j] + m[j, i]) * h
        v[k] = (m[k, i] + m[i, k]) * h
        w = (m[k, j] - m[j, k]) * h
        x, y, z = v
    return normalize(np.array([x, y, z, w]))
