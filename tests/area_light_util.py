"""Closed form for the cube area-light extension: the radiance a camera sees on a Lambertian floor point lit by a box
that emits uniformly from all its faces, with direct light only (max_depth 2).

Irradiance of a point p with normal n from a polygon of uniform radiance L (Lambert's formula): E = L/2 * |sum over
edges of theta_k * (gamma_k . n)|, theta_k the angle the edge subtends at p, gamma_k the unit normal of the plane
through p and the edge.  Radiance leaving a Lambertian surface of albedo rho: rho / pi * E.  A face contributes if p
is on its outer side; the floor is below the whole box, so every visible face is entirely above the horizon."""
import numpy as np

LIGHT_P0 = np.array([-1.0, 3.0, -1.5])
LIGHT_P1 = np.array([1.0, 3.25, 1.5])
LIGHT_RADIANCE = np.array([4.0, 3.0, 2.0])
FLOOR_ALBEDO = 0.5


def _faces(p0, p1):
    """(outward normal, 4 corners) of the six faces of the box"""
    out = []
    for k in range(3):
        i, j = (k + 1) % 3, (k + 2) % 3
        for side, plane in ((-1.0, p0[k]), (1.0, p1[k])):
            n = np.zeros(3)
            n[k] = side
            corners = []
            for a, b in ((p0[i], p0[j]), (p1[i], p0[j]), (p1[i], p1[j]), (p0[i], p1[j])):
                c = np.zeros(3)
                c[k], c[i], c[j] = plane, a, b
                corners.append(c)
            out.append((n, corners))
    return out


def polygon_irradiance(p, n, corners):
    """irradiance at p (normal n) per unit radiance of the polygon"""
    v = [c - p for c in corners]
    total = 0.0
    for k in range(len(v)):
        a, b = v[k], v[(k + 1) % len(v)]
        cr = np.cross(a, b)
        ln = np.linalg.norm(cr)
        if ln == 0.0:
            continue
        theta = np.arctan2(ln, np.dot(a, b))
        total += theta * np.dot(cr / ln, n)
    return abs(total) / 2.0


def floor_radiance(p, p0=LIGHT_P0, p1=LIGHT_P1, radiance=LIGHT_RADIANCE, albedo=FLOOR_ALBEDO):
    """RGB radiance leaving the floor point p (normal +y) towards anywhere, direct light of the box only"""
    n = np.array([0.0, 1.0, 0.0])
    e = 0.0
    for fn, corners in _faces(p0, p1):
        if np.dot(p - corners[0], fn) > 0.0:  # p on the outer side of the face: it sees the face
            e += polygon_irradiance(p, n, corners)
    return albedo / np.pi * e * radiance
