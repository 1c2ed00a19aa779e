"""Oracle robust loss (numpy fp64).  Test infrastructure.

Restates redescending_loss of src/build.py:382-395 (called as
misc.redescending_loss(w*slack, 3, 10, 20) at src/all_optimizations.py:497).
"""
import numpy as np


def _step(start, x):
    return 1.0 / (1.0 + np.exp(-(x - start)))


def redescending_loss(err, a, b, c):
    """rho(err): logistic blend of quadratic / linear / redescending / constant pieces."""
    e = np.abs(np.asarray(err, dtype=np.float64))
    sa, sb, sc = _step(a, e), _step(b, e), _step(c, e)
    cost = (1 - sa) / 2 * e ** 2
    cost = cost + (sa - sb) * (a * e - (a ** 2) / 2)
    cost = cost + (sb - sc) * (a * b - (a ** 2) / 2 + (a * (c - b) / 2) * (1 - ((c - e) / (c - b)) ** 2))
    cost = cost + sc * (a * b - (a ** 2) / 2 + (a * (c - b) / 2))
    return cost


def redescending_dloss(err, a, b, c):
    """(rho, d rho/d|err|, h) with h the Gauss-Newton curvature weight (see below)."""
    e = np.abs(np.asarray(err, dtype=np.float64))
    sa, sb, sc = _step(a, e), _step(b, e), _step(c, e)
    dsa, dsb, dsc = sa * (1 - sa), sb * (1 - sb), sc * (1 - sc)
    t2 = a * e - a * a / 2
    cb = c - b
    t3 = a * b - a * a / 2 + (a * cb / 2) * (1 - ((c - e) / cb) ** 2)
    t4 = a * b - a * a / 2 + a * cb / 2
    rho = (1 - sa) / 2 * e * e + (sa - sb) * t2 + (sb - sc) * t3 + sc * t4
    drho = (-dsa / 2 * e * e + (1 - sa) * e + (dsa - dsb) * t2 + (sa - sb) * a
            + (dsb - dsc) * t3 + (sb - sc) * (a * (c - e) / cb) + dsc * t4)
    # Gauss-Newton curvature weight: secant of rho' through its value at 0+ (rho has a small concave
    # kink at 0: rho'(0+) = d0 < 0), clipped to [0, 1].  Equals ~rho'' in the quadratic core, a/e in
    # the linear zone, ~0 beyond c.
    d0 = drho_at_zero(a, b, c)
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.where(e > 1e-12, (drho - d0) / np.maximum(e, 1e-300), 1.0)
    h = np.clip(h, 0.0, 1.0)
    return rho, drho, h


def drho_at_zero(a, b, c):
    """d rho / d e at e = 0+ (constant of the loss shape)."""
    sa, sb, sc = _step(a, 0.0), _step(b, 0.0), _step(c, 0.0)
    dsa, dsb, dsc = sa * (1 - sa), sb * (1 - sb), sc * (1 - sc)
    cb = c - b
    t2 = -a * a / 2
    t3 = a * b - a * a / 2 + (a * cb / 2) * (1 - (c / cb) ** 2)
    t4 = a * b - a * a / 2 + a * cb / 2
    return (dsa - dsb) * t2 + (sa - sb) * a + (dsb - dsc) * t3 + (sb - sc) * (a * c / cb) + dsc * t4
