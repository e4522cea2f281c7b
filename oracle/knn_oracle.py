"""TEST INFRASTRUCTURE: exact reference for simple_knn.distCUDA2 -- mean of the squared distances to the 3 nearest
neighbours, by scipy's k-d tree in float64.  (The simple-knn sources are not vendored under /root/reference, so there is
nothing to pin against; the definition is the one scene/gaussian_model.py:315 relies on and the library's README states.)"""
import numpy as np
from scipy.spatial import cKDTree


def mean_dist2_3nn(points):
    p = np.asarray(points, dtype=np.float64)
    n = p.shape[0]
    k = min(4, n)
    d, _ = cKDTree(p).query(p, k=k)          # first hit is the point itself (distance 0)
    d2 = np.square(d[:, 1:]) if k > 1 else np.zeros((n, 0))
    out = np.full(n, np.inf)
    if d2.shape[1] == 3:
        out = d2.sum(1) / 3.0
    return out
