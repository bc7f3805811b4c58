import numpy as np


def rel_l2(a, b):
    """relative L2 error of a against reference b (per problem when 2-D)"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.ndim == 1:
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)
