"""Noise assignment of the DNPH step (reference train/DNPH_TOMM/b_reg.py:5-40).  The Hungarian assignment is HOST
work upstream (scipy) and stays host work here (not shardable, O(B^3) on [B,B])."""
import numpy as np
from scipy.optimize import linear_sum_assignment


def rand_unit_rect(npoints, ndim):
    vec = np.random.randint(0, 2, size=(npoints, ndim))
    vec[vec == 0] = -1
    return vec


def gene_noise(embeedings, noises):
    """Assign each sample the +-1 noise row that minimises the total L2 cost (float64 like upstream)."""
    e = np.asarray(embeedings, dtype=np.float64)
    nz = np.asarray(noises, dtype=np.float64)
    losses = np.linalg.norm(e[:, None, :] - nz[None, :, :], axis=2)
    row_ind, col_ind = linear_sum_assignment(losses)
    new_noise = np.empty(shape=nz.shape, dtype='float64')
    new_noise[row_ind] = nz[col_ind]
    return new_noise
