"""Hamming ranking + mAP on the GPU (C ABI cmh_pack_* / cmh_hamming_map / cmh_hamming_dist) vs the goldens
produced by the reference's calc_map_k_matrix and vs the C++ oracle: per-query AP, mAP, and the full
ranking permutation bit-exact (tie order included)."""
import hashlib

import numpy as np
import pytest
import torch

import oracle
from maputil import CASES, case_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gpu_map(qB, rB, qL, rL, k=None, want_perm=False, depth_limit=-1, nq=None):
    import cmh_native as N
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    if nq is not None:
        qB, qL = qB[:nq], qL[:nq]
    return N.hamming_map(N.pack_codes(t(qB)), N.pack_labels(t(qL)), N.pack_codes(t(rB)), N.pack_labels(t(rL)),
                         rB.shape[1], rL.shape[1], topk=k, want_perm=want_perm, depth_limit=depth_limit)


@pytest.mark.parametrize("name", CASES)
def test_map_matches_reference_goldens(golden, name):
    g = golden("map.npz")
    qB, rB, qL, rL, k = case_inputs(g, name)
    N = rB.shape[0]
    mp, ap, _ = _gpu_map(qB, rB, qL, rL, k)
    np.testing.assert_allclose(ap.cpu().numpy(), g[f"{name}_ap"], rtol=0, atol=2e-6)
    assert abs(float(mp) - float(g[f"{name}_map"])) < 2e-6
    # bit-exact Hamming ranks for the recorded queries
    nperm = len(g[f"{name}_ind_sha"])
    _, _, perm = _gpu_map(qB, rB, qL, rL, k, want_perm=True, nq=nperm)
    perm = perm.cpu().numpy()
    tsum = (qL[:nperm] @ rL.T > 0).sum(1)
    for i in range(nperm):
        if tsum[i] == 0:
            assert (perm[i] == -1).all()       # skipped query (utils/calc_utils.py:27-29)
            continue
        assert hashlib.sha256(perm[i].astype(np.int64).tobytes()).hexdigest() == str(g[f"{name}_ind_sha"][i])
        if N <= 5003:
            assert np.array_equal(perm[i], g[f"{name}_ind{i}"])


@pytest.mark.parametrize("n,K", [(1, 16), (77, 64), (1000, 128), (513, 48)])
def test_unpack_codes_is_the_inverse_of_pack_codes(n, K):
    """cmh_unpack_codes: what data-parallel evaluation does with the gathered bit planes (train/base.py::_gather_code_shards)."""
    import cmh_native as N
    g = torch.Generator().manual_seed(n + K)
    codes = torch.randint(-1, 2, (n, K), generator=g).float().to(DEV)          # -1, 0, +1: sign() yields exact zeros too
    sp, nz = N.pack_codes(codes)
    assert torch.equal(N.unpack_codes(sp, nz, K), codes)
    with pytest.raises(N.NativeError):
        N.unpack_codes(sp, nz, K + 64)


def test_map_mean_is_the_ranking_kernels_own_mean(golden):
    """cmh_map_mean on a per-query AP vector = the mAP cmh_hamming_map returns with it, bit for bit (dist_utils.mean_in_query_order
    uses it on APs gathered from query shards)."""
    import cmh_native as N
    g = golden("map.npz")
    qB, rB, qL, rL, k = case_inputs(g, "corr_1k_64_k50")
    mp, ap, _ = _gpu_map(qB, rB, qL, rL, k)
    assert float(N.map_mean(ap)) == float(mp)
    acc = np.float32(0)
    for v in ap.cpu().numpy():
        acc = np.float32(acc + v)
    assert float(N.map_mean(ap)) == float(acc / np.float32(ap.numel()))
    import dist_utils as du
    assert float(du.mean_in_query_order(ap)) == float(mp)


def test_entry_point_calc_map_k_matrix(golden):
    """The reference-shaped API (utils/calc_utils.py) with CPU labels, like train/base.py:259 passes them."""
    from utils.calc_utils import calc_hammingDist, calc_map_k_matrix, calc_neighbor
    g = golden("map.npz")
    qB, rB, qL, rL, k = case_inputs(g, "corr_1k_64_k50")
    m = calc_map_k_matrix(torch.from_numpy(qB).to(DEV), torch.from_numpy(rB).to(DEV), torch.from_numpy(qL),
                          torch.from_numpy(rL), k, 0)
    assert m.dim() == 0 and abs(float(m) - float(g["corr_1k_64_k50_map"])) < 2e-6
    d = calc_hammingDist(torch.from_numpy(qB[3]).to(DEV), torch.from_numpy(rB).to(DEV))
    assert d.shape == (1, rB.shape[0])
    np.testing.assert_array_equal(d.cpu().numpy()[0], oracle.hamming_row(qB[3], rB))
    s = calc_neighbor(torch.from_numpy(qL[:50]).to(DEV), torch.from_numpy(rL[:70]).to(DEV))
    np.testing.assert_array_equal(s.cpu().numpy(), (qL[:50] @ rL[:70].T > 0).astype(np.float32))


def test_codes_with_zeros_and_half_integer_distances(golden):
    g = golden("map.npz")
    qB, rB, qL, rL, k = case_inputs(g, "zeros_300_32")
    assert (qB == 0).any() and (rB == 0).any()
    import cmh_native as N
    t = lambda a: torch.from_numpy(a).to(DEV)
    d = N.hamming_dist(N.pack_codes(t(qB)), N.pack_codes(t(rB)), 32).cpu().numpy()
    ref = np.stack([oracle.hamming_row(q, rB) for q in qB])
    np.testing.assert_array_equal(d, ref)
    assert (d % 1 == 0.5).any()


@pytest.mark.parametrize("depth", [0, 1, 3, 6])
def test_heapsort_fallback_matches_libstdcpp(depth):
    """Force the introsort depth budget so the __partial_sort (heapsort) branch runs on the GPU."""
    rng = np.random.default_rng(depth)
    Q, N, K, C = 6, 3000, 32, 8
    qB = np.where(rng.random((Q, K)) < 0.5, -1.0, 1.0).astype(np.float32)
    rB = np.where(rng.random((N, K)) < 0.5, -1.0, 1.0).astype(np.float32)
    qL = np.ones((Q, C), np.float32)
    rL = (rng.random((N, C)) < 0.3).astype(np.float32)
    _, _, perm = _gpu_map(qB, rB, qL, rL, want_perm=True, depth_limit=depth)
    perm = perm.cpu().numpy()
    for i in range(Q):
        ref = oracle.sort_perm_depth(oracle.hamming_row(qB[i], rB), depth)
        assert np.array_equal(perm[i], ref), (depth, i)


@pytest.mark.parametrize("K,N,depth", [(8, 6000, -1), (16, 9000, -1), (8, 5000, 7), (8, 5000, 11), (4, 2500, -1), (32, 20000, -1)])
def test_long_runs_of_equal_keys(K, N, depth):
    """Few key values -> runs of hundreds of equal keys: the all-equal subtrees are applied as one closed-form permutation
    (csrc/hamming_map.hip::all_equal_shortcut), with a depth budget that sometimes forbids it; N = 20000 takes the
    workspace (non-LDS) path."""
    rng = np.random.default_rng(K * 1000 + N + depth)
    Q, C = 5, 6
    qB = np.where(rng.random((Q, K)) < 0.5, -1.0, 1.0).astype(np.float32)
    rB = np.where(rng.random((N, K)) < 0.5, -1.0, 1.0).astype(np.float32)
    rB[: N // 3] = rB[0]                                   # one third of the database shares a code
    qL = np.ones((Q, C), np.float32)
    rL = (rng.random((N, C)) < 0.3).astype(np.float32)
    _, _, perm = _gpu_map(qB, rB, qL, rL, want_perm=True, depth_limit=depth)
    perm = perm.cpu().numpy()
    for i in range(Q):
        keys = oracle.hamming_row(qB[i], rB)
        ref = oracle.sort_perm_depth(keys, depth) if depth >= 0 else oracle.sort_perm(keys)
        assert np.array_equal(perm[i], ref), (K, N, depth, i)


def test_size_independent_properties_at_full_size():
    """NUS-WIDE scale (N=190 834, 128-bit): ranking is a permutation, keys are sorted, AP in [0,1], identical
    queries give identical APs, and a query that equals a relevant DB code ranks a relevant item first."""
    import cmh_native as N
    rng = np.random.default_rng(5)
    Q, Nn, K, C = 8, 190834, 128, 21
    rL = (rng.random((Nn, C)) < 0.15).astype(np.float32)
    rB = np.where(rng.random((Nn, K)) < 0.5, -1.0, 1.0).astype(np.float32)
    qB = rB[:Q].copy()
    qL = rL[:Q].copy()
    qL[qL.sum(1) == 0, 0] = 1
    rL[:Q] = qL
    qB[1], qL[1] = qB[0], qL[0]
    mp, ap, perm = _gpu_map(qB, rB, qL, rL, want_perm=True)
    ap, perm = ap.cpu().numpy(), perm.cpu().numpy()
    assert ((ap >= 0) & (ap <= 1)).all() and ap[0] == ap[1]
    for i in range(Q):
        assert np.array_equal(np.sort(perm[i]), np.arange(Nn))
        keys = oracle.hamming_row(qB[i], rB)[perm[i]]
        assert (np.diff(keys) >= 0).all()
        assert keys[0] == 0
    m_ref, ap_ref, _ = oracle.map_k(qB, rB, qL, rL)
    np.testing.assert_allclose(ap, ap_ref, rtol=0, atol=2e-6)


def test_bad_codes_are_refused():
    import cmh_native as N
    c = torch.ones(4, 16, device=DEV)
    c[2, 3] = 0.5
    with pytest.raises(N.NativeError):
        N.pack_codes(c)
    with pytest.raises(N.NativeError):
        N.pack_labels(-torch.ones(2, 5, device=DEV))


@pytest.mark.parametrize("Q,N,K,C,k,zeros", [(5, 7, 16, 4, None, False), (9, 200, 64, 8, 50, False), (6, 1000, 128, 12, None, True),
                                              (4, 4097, 512, 24, 1000, False), (3, 5000, 2048, 24, None, True),
                                              (3, 30000, 64, 24, 5000, False), (3, 12000, 64, 24, None, False)])   # 12 000: elements in LDS, lists in the workspace
def test_stable_tie_order_matches_stable_sort(Q, N, K, C, k, zeros):
    """CMH_TIE_STABLE: ties by ascending database index == the oracle's std::stable_sort ranking, bit-exact (1, 2 and 3
    radix passes; LDS- and workspace-resident queries; codes with zeros; a query without relevant items)."""
    import cmh_native as Nn
    rng = np.random.default_rng(Q * 1000 + N + K)
    vals = np.array([-1.0, 1.0, 0.0] if zeros else [-1.0, 1.0], np.float32)
    qB = vals[rng.integers(0, len(vals), (Q, K))]
    rB = vals[rng.integers(0, len(vals), (N, K))]
    qL = (rng.random((Q, C)) < 0.2).astype(np.float32)
    rL = (rng.random((N, C)) < 0.2).astype(np.float32)
    qL[0] = 0                                             # no relevant item: skipped by the reference
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    mp, ap, perm = Nn.hamming_map(Nn.pack_codes(t(qB)), Nn.pack_labels(t(qL)), Nn.pack_codes(t(rB)), Nn.pack_labels(t(rL)),
                                  K, C, topk=k, tie_order=Nn.TIE_STABLE, want_perm=True)
    m_ref, ap_ref, ind = oracle.map_k(qB, rB, qL, rL, k, stable=True, want_ind=True)
    perm = perm.cpu().numpy()
    assert (perm[0] == -1).all()
    np.testing.assert_array_equal(perm[1:], ind[1:])
    np.testing.assert_allclose(ap.cpu().numpy(), ap_ref, rtol=0, atol=2e-6)
    assert abs(float(mp) - float(m_ref)) < 2e-6
    # and it is NOT the reference's order whenever ties exist (the two modes must not be confused)
    if N >= 1000 and K <= 128:
        _, _, perm_ref = Nn.hamming_map(Nn.pack_codes(t(qB)), Nn.pack_labels(t(qL)), Nn.pack_codes(t(rB)), Nn.pack_labels(t(rL)),
                                        K, C, topk=k, want_perm=True)
        assert not np.array_equal(perm_ref.cpu().numpy()[1:], perm[1:])


@pytest.mark.parametrize("N", [7512, 7513, 16276, 16277, 33027, 33028])
def test_ranking_at_the_placement_boundaries(N):
    """csrc/hamming_map.hip keeps a query's working set in LDS (twice per CU up to 7 512 items), elements in LDS + lists in the workspace
    (twice per CU up to 16 276, once up to 33 027) or all of it in the workspace: the last size of each placement and the first of
    the next, complete permutations against the oracle's introsort and the stable order against std::stable_sort."""
    import cmh_native as Nn
    rng = np.random.default_rng(N)
    Q, K, C = 3, 32, 8
    qB = np.where(rng.random((Q, K)) < 0.5, -1.0, 1.0).astype(np.float32)
    rB = np.where(rng.random((N, K)) < 0.5, -1.0, 1.0).astype(np.float32)
    qL = (rng.random((Q, C)) < 0.3).astype(np.float32)
    rL = (rng.random((N, C)) < 0.3).astype(np.float32)
    qL[:, 0] = 1.0                      # (a query without any relevant item is skipped like upstream's `continue`: no ranking to compare)
    rL[:7, 0] = 1.0
    _, ap, perm = _gpu_map(qB, rB, qL, rL, want_perm=True)
    perm = perm.cpu().numpy()
    for i in range(Q):
        assert np.array_equal(perm[i], oracle.sort_perm(oracle.hamming_row(qB[i], rB))), (N, i)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    _, _, sperm = Nn.hamming_map(Nn.pack_codes(t(qB)), Nn.pack_labels(t(qL)), Nn.pack_codes(t(rB)), Nn.pack_labels(t(rL)), K, C,
                                 tie_order=Nn.TIE_STABLE, want_perm=True)
    for i in range(Q):
        keys = oracle.hamming_row(qB[i], rB)
        assert np.array_equal(sperm[i].cpu().numpy(), np.argsort(keys, kind="stable")), (N, i)
