"""The closed-form / breadth-first introsort decomposition used by the GPU kernel equals libstdc++
std::sort (the reference's torch.sort tie order) — checked on the CPU against the oracle."""
import numpy as np
import pytest

import oracle
from introsort_model import emulate


def _keys(rng, N, nk, kind):
    if kind == 0:
        return rng.integers(0, nk, N)
    if kind == 1:
        return np.clip(np.round(rng.normal(nk / 2, nk / 8 + 0.5, N)), 0, nk).astype(int)
    k = np.sort(rng.integers(0, nk, N))
    return k if kind == 2 else k[::-1].copy()


@pytest.mark.parametrize("seed", range(4))
def test_model_equals_std_sort(seed):
    rng = np.random.default_rng(seed)
    for trial in range(60):
        N = int(rng.integers(1, 2500))
        nk = int(rng.choice([1, 2, 3, 5, 17, 33, 129, 257]))
        keys = _keys(rng, N, nk, trial % 4)
        assert np.array_equal(emulate(keys), oracle.sort_perm(keys.astype(np.float32))), (seed, trial, N, nk)


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 5])
def test_model_heapsort_fallback(depth):
    rng = np.random.default_rng(100 + depth)
    for trial in range(25):
        N = int(rng.integers(17, 1500))
        nk = int(rng.choice([2, 5, 33, 129]))
        keys = _keys(rng, N, nk, trial % 4)
        ref = oracle.sort_perm_depth(keys.astype(np.float32), depth)
        assert np.array_equal(emulate(keys, depth_override=depth), ref), (depth, trial, N, nk)
