"""Shared helpers of the parity tests."""
import numpy as np


def rel_fro(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# tolerances stated by BASELINE.json north_star
TOL_P = 1e-6     # relative Frobenius on P
TOL_DX = 1e-8    # relative 2-norm on dx
