"""The loader / consumer GEMM kernel (round 5; csrc/gemm_lc.hip, include/cmh.h: cmh_set_gemm_lc).

Four waves of a workgroup stage operands, four multiply: another schedule of the SAME arithmetic as the wide kernel (same MFMA chain
over K per output element, same epilogue order, the same residual-first rule), so every comparison is torch.equal against the wide
kernel - which tests/test_gpu_kernels.py pins to fp64 and, through the towers, to the reference's goldens (model/base/model.py:171-196:
the four nn.Linear of a ResidualAttentionBlock)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(shape, g, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


def _problem(M, Nn, K, kind, g):
    p = {"x": _rand((M, K), g).bfloat16().to(DEV), "w": _rand((Nn, K), g, K ** -0.5).bfloat16().to(DEV), "bias": _rand((Nn,), g).to(DEV)}
    if kind == 1:
        p["residual"] = _rand((M, Nn), g).half().to(DEV)
    return p


def _plain(N, p, kind):
    return N.linear_gemm(p["x"], p["w"], bias=p["bias"], residual=p.get("residual"), quickgelu=kind == 2, out_bf16=kind != 1, out_f16=kind == 1)


# (M, N, K, kind): kind 0 bias -> bf16, 1 bias + fp16 residual -> fp16 (K <= 1024: the residual comes first), 2 bias + QuickGELU -> bf16
PLAIN = [
    (12800, 2304, 768, 0), (12800, 768, 768, 1), (12800, 3072, 768, 2), (12800, 768, 3072, 1),       # the image tower's block, batch 256
    (10499, 1536, 512, 0), (10499, 512, 512, 1), (10499, 2048, 512, 2), (10499, 512, 2048, 1),       # packed text rows
    (2049, 256, 256, 0),          # 4 K-steps (the shortest tile the kernel takes), 17 row tiles, the last one a single row
    (130, 512, 1024, 1),          # fewer tiles than an XCD has workgroups; 16 K-steps: the residual still comes first
    (300, 256, 1088, 1),          # 17 K-steps: the residual behind the bias
    (5000, 1024, 576, 2),         # K not a multiple of 128: 9 K-steps
]


@pytest.mark.parametrize("case", range(len(PLAIN)))
def test_lc_kernel_gives_the_wide_kernels_bits(case):
    import cmh_native as N
    M, Nn, K, kind = PLAIN[case]
    g = torch.Generator().manual_seed(500 + case)
    p = _problem(M, Nn, K, kind, g)
    try:
        N.set_gemm_rows(0)                       # (few-row launches would leave both kernels for the 64 x 64 one)
        N.set_gemm_lc(0)
        ref = _plain(N, p, kind)
        N.set_gemm_lc(1)
        got = _plain(N, p, kind)
    finally:
        N.set_gemm_lc(-1)
        N.set_gemm_rows(-1)
    assert torch.equal(ref, got)


GROUPED = [
    ((12800, 2304, 768), (10499, 1536, 512), 0),
    ((12800, 768, 768), (10499, 512, 512), 1),
    ((12800, 3072, 768), (10499, 2048, 512), 2),
    ((12800, 768, 3072), (10499, 512, 2048), 1),
    ((2100, 256, 512), (4000, 1024, 256), 0),      # 'a' shorter in K than 'b': the launcher swaps them
    ((2049, 512, 1024), (2500, 256, 1024), 1),     # a handful of tiles each; some workgroups own tiles of the second problem only
    ((9000, 1024, 512), (4100, 512, 1024), 2),
]


@pytest.mark.parametrize("case", range(len(GROUPED)))
def test_lc_grouped_launch_gives_the_wide_kernels_bits(case, monkeypatch):
    import cmh_native as N
    (Ma, Na, Ka), (Mb, Nb, Kb), kind = GROUPED[case]
    g = torch.Generator().manual_seed(600 + case)
    probs = [_problem(Ma, Na, Ka, kind, g), _problem(Mb, Nb, Kb, kind, g)]
    out = "f16" if kind == 1 else "bf16"
    md = torch.tensor([Mb - 37], dtype=torch.int32, device=DEV)      # a device-side row count (packed captions)
    try:
        N.set_gemm_lc(0)
        ref = [_plain(N, p, kind) for p in probs]
        N.set_gemm_lc(1)
        got = N.linear_gemm_grouped(probs, quickgelu=kind == 2, out=out)
        got_md = N.linear_gemm_grouped(probs, quickgelu=kind == 2, out=out, m_dev=(None, md))
    finally:
        N.set_gemm_lc(-1)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    assert torch.equal(got_md[0], ref[0]) and torch.equal(got_md[1][:Mb - 37], ref[1][:Mb - 37])


def test_lc_kernel_never_writes_rows_past_the_device_side_count():
    import ctypes as C
    import cmh_native as N
    g = torch.Generator().manual_seed(7)
    M, Nn, K = 3000, 512, 512
    probs = [_problem(M, Nn, K, 0, g), _problem(M, Nn, K, 0, g)]
    md = torch.tensor([1234], dtype=torch.int32, device=DEV)
    outs = [torch.full((M, Nn), -7.0, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    structs = [N.GemmProblem(N.ptr(p["x"]), N.ptr(p["w"]), N.ptr(p["bias"]), N.ptr(None), N.ptr(o), M, Nn, K, N.ptr(m), N.ptr(None), 1.0, 1.0)
               for p, o, m in zip(probs, outs, (None, md))]
    try:
        N.set_gemm_lc(1)
        N.check(N.lib().cmh_linear_gemm_grouped(N.BF16, C.byref(structs[0]), C.byref(structs[1]), N.EPI_BIAS | N.EPI_OUT_BF16,
                                                N.stream_ptr(torch.device(DEV))), "cmh_linear_gemm_grouped")
        torch.cuda.synchronize()
    finally:
        N.set_gemm_lc(-1)
    assert bool((outs[1][1234:] == -7.0).all()) and not bool((outs[1][:1234] == -7.0).all()) and not bool((outs[0] == -7.0).any())


# ---- the 12-wave form (cmh_set_gemm_lc(4)): 4 staging waves + 8 MFMA waves of 64 x 64, stores deferred into the next tile's K-steps.
# It takes every block launch with K >= 512 (bias, + QuickGELU, + fp16 residual first / behind the bias).
LC2_PLAIN = [(12800, 2304, 768, 0), (12800, 3072, 768, 2), (10499, 1536, 512, 0), (10499, 2048, 512, 2),
             (12800, 768, 768, 1), (12800, 768, 3072, 1), (10499, 512, 512, 1), (10499, 512, 2048, 1),      # the residual launches: first (K <= 1024) / behind the bias
             (300, 256, 1088, 1), (2049, 512, 1024, 1),
             (2049, 256, 512, 0),          # 8 K-steps (the shortest tile it takes), the last row tile a single row
             (5000, 1024, 576, 2),         # 9 K-steps
             (130, 512, 1024, 0)]          # fewer tiles than an XCD has workgroups: every tile is its workgroup's last (direct stores)


@pytest.mark.parametrize("case", range(len(LC2_PLAIN)))
def test_lc2_kernel_gives_the_wide_kernels_bits(case):
    import cmh_native as N
    M, Nn, K, kind = LC2_PLAIN[case]
    g = torch.Generator().manual_seed(700 + case)
    p = _problem(M, Nn, K, kind, g)
    try:
        N.set_gemm_rows(0)
        N.set_gemm_lc(0)
        ref = _plain(N, p, kind)
        N.set_gemm_lc(4)
        N.prof_gemm_begin(8)
        got = _plain(N, p, kind)
        N.prof_gemm_end()
    finally:
        N.set_gemm_lc(-1)
        N.set_gemm_rows(-1)
    assert torch.equal(ref, got)


@pytest.mark.parametrize("case", range(len(GROUPED)))
def test_lc2_grouped_launch_gives_the_wide_kernels_bits(case):
    import cmh_native as N
    (Ma, Na, Ka), (Mb, Nb, Kb), kind = GROUPED[case] if case != 4 else ((2100, 256, 512), (4000, 1024, 1024), 0)
    g = torch.Generator().manual_seed(800 + case)
    probs = [_problem(Ma, Na, Ka, kind, g), _problem(Mb, Nb, Kb, kind, g)]
    md = torch.tensor([Mb - 37], dtype=torch.int32, device=DEV)
    try:
        N.set_gemm_lc(0)
        ref = [_plain(N, p, kind) for p in probs]
        N.set_gemm_lc(4)
        out = "f16" if kind == 1 else "bf16"
        got = N.linear_gemm_grouped(probs, quickgelu=kind == 2, out=out)
        got_md = N.linear_gemm_grouped(probs, quickgelu=kind == 2, out=out, m_dev=(None, md))
    finally:
        N.set_gemm_lc(-1)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    assert torch.equal(got_md[0], ref[0]) and torch.equal(got_md[1][:Mb - 37], ref[1][:Mb - 37])


# ---- the 12-wave form on e4m3 operands (cmh_set_gemm_lc(7)): bias + dequantisation, bf16 output - the fp8 mode's QKV launches
@pytest.mark.parametrize("M,Nn,K", [(12800, 2304, 768), (10499, 1536, 512), (2049, 256, 512), (130, 512, 1024)])
def test_lc2q_kernel_gives_the_wide_fp8_kernels_bits(M, Nn, K):
    import cmh_native as N
    g = torch.Generator().manual_seed(900 + M)
    e4m3 = lambda shape, sc: (torch.randn(*shape, generator=g) * sc).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    x, w = e4m3((M, K), 1.0), e4m3((Nn, K), 8 * K ** -0.5)
    cs, b = (torch.rand(Nn, generator=g) * 0.1 + 0.05).to(DEV), _rand((Nn,), g).to(DEV)
    try:
        N.set_gemm_rows(0)
        N.set_gemm_lc(0)
        ref = N.linear_gemm_fp8(x, w, cs, 0.37, bias=b, out="bf16")
        N.set_gemm_lc(7)
        got = N.linear_gemm_fp8(x, w, cs, 0.37, bias=b, out="bf16")
        other = N.linear_gemm_fp8(x, w, cs, 0.37, bias=b, out="f16")      # a form it does not take: the wide kernel runs
        N.set_gemm_lc(0)
        other_ref = N.linear_gemm_fp8(x, w, cs, 0.37, bias=b, out="f16")
    finally:
        N.set_gemm_lc(-1)
        N.set_gemm_rows(-1)
    assert torch.equal(ref, got) and torch.equal(other, other_ref)
