"""Comparison rules for SVD factors (sign ambiguity, gaps) used by the tests."""
import numpy as np

EPS32 = float(np.finfo(np.float32).eps)


def align_signs(U, Uref):
    """Flip columns of U so that <u_i, uref_i> >= 0; returns the signs."""
    sg = np.sign(np.sum(U * Uref, axis=0))
    sg[sg == 0] = 1
    return sg


def col_cosines(U, Uref):
    nu = np.linalg.norm(U, axis=0) * np.linalg.norm(Uref, axis=0)
    return np.abs(np.sum(U * Uref, axis=0)) / nu


def sv_tolerance(s_ref, c=64.0):
    """Stated fp32 tolerance of the Gram route: |ds_i| <= c * eps32 * s_1^2 / s_i."""
    s_ref = np.asarray(s_ref, dtype=np.float64)
    return c * EPS32 * s_ref[0] ** 2 / s_ref


def vec_tolerance(s_ref, c=64.0):
    """sin(angle) bound for vector i: c * eps32 * s_1^2 / min gap of s^2 to its neighbours."""
    s2 = np.asarray(s_ref, dtype=np.float64) ** 2
    gap = np.empty_like(s2)
    for i in range(len(s2)):
        g = []
        if i > 0:
            g.append(s2[i - 1] - s2[i])
        if i + 1 < len(s2):
            g.append(s2[i] - s2[i + 1])
        gap[i] = max(min(g), 1e-300) if g else s2[i]
    return c * EPS32 * s2[0] / gap
