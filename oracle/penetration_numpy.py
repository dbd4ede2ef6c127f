"""ORACLE (test infrastructure): a SECOND, independent statement of the interpenetration term, for cross-checking
oracle/penetration.py (PARITY UNPINNED: the package the reference calls is absent; SURVEY.md appendix A asks for two independent
implementations wherever an external algorithm is restated).

Plain loops over triangle pairs in float64, no torch, no vectorisation, and different routes to the same quantities:
  * candidates: a double loop over all triangle pairs with explicit interval tests and set intersection of vertex ids;
  * the circumscribed circle by solving the 2 x 2 system |c - p0|^2 = |c - p1|^2 = |c - p2|^2 in the triangle's plane basis
    (penetration.py uses the closed cross-product formula);
  * the height / radial decomposition through an orthonormal frame of the plane (penetration.py subtracts the normal part);
  * the part rules written out as in fit_single_frame.py:318-328's description.
Formulas: Tzionas et al. IJCV 2016 eqs. 13-16 as summarised in the header of penetration.py (assumptions A1-A7 apply alike)."""
import numpy as np


def colliding_pairs(verts, faces, segm=None, parents=None, ign_pairs=()):
    verts = np.asarray(verts, np.float64)
    faces = np.asarray(faces, np.int64)
    ign = set((min(a, b), max(a, b)) for a, b in ign_pairs)
    out = []
    boxes = [(verts[f].min(0), verts[f].max(0)) for f in faces]
    for i in range(len(faces)):
        for j in range(i + 1, len(faces)):
            (la, ha), (lb, hb) = boxes[i], boxes[j]
            if any(la[e] > hb[e] or lb[e] > ha[e] for e in range(3)):
                continue
            if set(faces[i].tolist()) & set(faces[j].tolist()):
                continue
            if segm is not None:
                a, b = int(segm[i]), int(segm[j])
                if a == b or int(parents[j]) == a or int(parents[i]) == b or (min(a, b), max(a, b)) in ign:
                    continue
            out.append((i, j))
    return out


def _circumcircle(p0, p1, p2):
    e1 = p1 - p0
    n = np.cross(e1, p2 - p0)
    n = n / np.linalg.norm(n)
    u = e1 / np.linalg.norm(e1)
    w = np.cross(n, u)
    # plane coordinates of the three corners; centre from the perpendicular bisectors
    q1 = np.array([np.dot(p1 - p0, u), np.dot(p1 - p0, w)])
    q2 = np.array([np.dot(p2 - p0, u), np.dot(p2 - p0, w)])
    A = 2.0 * np.array([q1, q2])
    c2 = np.linalg.solve(A, np.array([q1 @ q1, q2 @ q2]))
    centre = p0 + c2[0] * u + c2[1] * w
    return centre, float(np.linalg.norm(c2)), n, u, w


def _upsilon(x, sigma):
    if x <= -sigma:
        return -x + 1.0 - sigma
    if x < sigma:
        return -(1.0 - 2.0 * sigma) / (4.0 * sigma ** 2) * x ** 2 - x / (2.0 * sigma) + (3.0 - 2.0 * sigma) / 4.0
    return 0.0


def _psi(tri, point, sigma, penalize_outside):
    o, r, n, u, w = _circumcircle(*tri)
    d = point - o
    x = float(np.dot(d, n))
    rho = float(np.hypot(np.dot(d, u), np.dot(d, w)))
    if not x < sigma or (not penalize_outside and x > 0.0):
        return 0.0
    phi = rho / (r - (r / sigma) * x)
    if not phi < 1.0:
        return 0.0
    return ((1.0 - phi) * _upsilon(x, sigma)) ** 2


def loss(verts, faces, pairs, sigma, penalize_outside=True, point2plane=False):
    verts = np.asarray(verts, np.float64)
    total = 0.0
    for i, j in pairs:
        ta, tb = verts[np.asarray(faces[i])], verts[np.asarray(faces[j])]
        s = sum(_psi(ta, p, sigma, penalize_outside) ** 2 for p in tb) + sum(_psi(tb, p, sigma, penalize_outside) ** 2 for p in ta)
        if point2plane:
            s *= float(np.dot(_circumcircle(*ta)[2], _circumcircle(*tb)[2])) ** 2
        total += s
    return total
