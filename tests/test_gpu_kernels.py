"""Unit-level parity of the tower building blocks, called through the C ABI (cmh_linear_gemm,
cmh_layernorm, cmh_attention) against a plain fp32/fp64 torch-CPU / numpy statement of the same op."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("M,N,K", [(1, 128, 64), (130, 128, 128), (256, 512, 768), (333, 384, 3072), (2000, 2304, 768)])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_gemm_epilogues(M, N, K, mode):
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    if mode == "bf16":
        xd, wd = x.bfloat16().to(_dev()), w.bfloat16().to(_dev())
        xr, wr = x.bfloat16().double(), w.bfloat16().double()      # same rounded operands
        tol = dict(rtol=2e-3, atol=2e-3)
    else:
        xd, wd = x.to(_dev()), w.to(_dev())
        xr, wr = x.double(), w.double()
        tol = dict(rtol=1e-5, atol=2e-5)
    base = xr @ wr.t()
    # plain
    out = Nn.linear_gemm(xd, wd)
    torch.testing.assert_close(out.cpu().double(), base, **tol)
    # bias + quickgelu -> bf16/f32 out
    v = base + b.double()
    ref = v * torch.sigmoid(1.702 * v)
    out = Nn.linear_gemm(xd, wd, bias=b.to(_dev()), quickgelu=True, out_bf16=(mode == "bf16"))
    t2 = dict(rtol=1e-2, atol=1e-2) if mode == "bf16" else tol
    torch.testing.assert_close(out.cpu().double(), ref, **t2)
    # bias + residual (in-place style)
    out = Nn.linear_gemm(xd, wd, bias=b.to(_dev()), residual=r.to(_dev()))
    torch.testing.assert_close(out.cpu().double(), base + b.double() + r.double(), **tol)


@pytest.mark.parametrize("M,N,K", [(12800, 3072, 768), (10499, 2048, 512), (6500, 1024, 1024)])
@pytest.mark.parametrize("out", ["bf16", "f16"])
def test_deferred_quickgelu_gives_the_epilogue_bits(M, N, K, out):
    """csrc/gemm_wide.hip, DGE: on the 128-row tile the QuickGELU of a tile runs between the NEXT tile's MFMAs (several tiles per
    workgroup at these sizes, the last one partial).  Same arithmetic per element as the 160- and 96-row tiles' epilogues: equal bits."""
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M + N + K)
    xd = torch.randn(M, K, generator=g).bfloat16().to(_dev())
    wd = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(_dev())
    bd = torch.randn(N, generator=g).to(_dev())
    outs = {}
    try:
        for rows in (128, 160, 96, -1):
            Nn.gemm_tuning(rows, -1)
            outs[rows] = Nn.linear_gemm(xd, wd, bias=bd, quickgelu=True, out_bf16=out == "bf16", out_f16=out == "f16")
    finally:
        Nn.gemm_tuning(-1, -1)
    v = xd[:64].double() @ wd.double().t() + bd.double()
    torch.testing.assert_close(outs[128][:64].double(), v * torch.sigmoid(1.702 * v), rtol=1e-2, atol=1e-2)
    for rows in (160, 96, -1):
        assert torch.equal(outs[128], outs[rows]), rows


@pytest.mark.parametrize("M,N,K", [(1000, 2304, 768), (10499, 512, 512), (3000, 1536, 512), (777, 768, 3072)])
def test_gemm_wide_tile_variants(M, N, K):
    """Every tile height (96 / 128 / 160 rows) and both tile orders of the wide kernel give the SAME bits (an output element is
    the same K-ordered MFMA chain whatever tile it falls in), and those bits match the fp64 statement of the op."""
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).half()
    ref_q = x.double() @ w.double().t() + b.double()
    ref_q = ref_q * torch.sigmoid(1.702 * ref_q)
    ref_r = x.double() @ w.double().t() + b.double() + r.double()
    xd, wd, bd, rd = (t.to(_dev()) for t in (x, w, b, r))
    first = None
    try:
        for rows in (-1, 96, 128, 160):
            for order in (-1, 0, 2):
                Nn.gemm_tuning(rows, order)
                oq = Nn.linear_gemm(xd, wd, bias=bd, quickgelu=True, out_bf16=True)
                orr = Nn.linear_gemm(xd, wd, bias=bd, residual=rd, out_f16=True)
                if first is None:
                    first = (oq, orr)
                    torch.testing.assert_close(oq.cpu().double(), ref_q, rtol=1e-2, atol=1e-2)
                    torch.testing.assert_close(orr.cpu().double(), ref_r, rtol=1.5e-3, atol=2e-3)
                else:
                    assert torch.equal(oq, first[0]), (rows, order)
                    assert torch.equal(orr, first[1]), (rows, order)
    finally:
        Nn.gemm_tuning(-1, -1)
    with pytest.raises(Nn.NativeError):
        Nn.gemm_tuning(100, -1)


@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (256, 3072, 768), (256, 768, 3072), (256, 512, 2048), (200, 512, 512), (37, 1536, 512), (512, 128, 64), (1600, 768, 768), (2048, 2304, 768)])
@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_gemm_rows_kernel_gives_the_wide_kernels_bits(M, N, K, mode):
    """csrc/gemm_rows.hip (few rows, 64 x 64 tiles) against the wide / 128 x 128 kernels on the same operands: identical bits
    for every epilogue it takes over (the pooled-row tail relies on it), and both equal the fp64 statement within the mode's tolerance."""
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    if mode == "bf16":
        x, w = x.bfloat16(), w.bfloat16()
    xd, wd, bd = x.to(_dev()), w.to(_dev()), b.to(_dev())
    wide_ok = N % 256 == 0
    cases = [dict(), dict(bias=bd, quickgelu=True, out_bf16=(mode == "bf16")), dict(bias=bd, residual=r.to(_dev()))]
    if wide_ok:
        cases.append(dict(bias=bd, residual=r.half().to(_dev()), out_f16=True))
    outs = {}
    try:
        for on in (1, 0):
            Nn.set_gemm_rows(on)
            outs[on] = [Nn.linear_gemm(xd, wd, **kw) for kw in cases]
    finally:
        Nn.set_gemm_rows(-1)
    for i, (a, c) in enumerate(zip(outs[1], outs[0])):
        if i > 0 and not wide_ok:
            # N % 256 != 0 goes to the 128 x 128 fallback kernels, which divide in QuickGELU and add the residual last: the rows
            # kernel follows the WIDE kernel's operation order (v_rcp form; residual first when K is short), so only close here
            torch.testing.assert_close(a.float(), c.float(), rtol=1e-2 if mode == "bf16" else 1e-5, atol=1e-2 if mode == "bf16" else 1e-5)
            continue
        assert torch.equal(a, c), (i, float((a.float() - c.float()).abs().max()))
    ref = x.double() @ w.double().t()
    tol = dict(rtol=2e-3, atol=2e-3) if mode == "bf16" else dict(rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(outs[1][0].cpu().double(), ref, **tol)
    torch.testing.assert_close(outs[1][2].cpu().double(), ref + b.double() + r.double(), **tol)


def test_gemm_asymmetric_identity():
    """A = I against an ASYMMETRIC W catches a transposed C write (cdna guide §3)."""
    import cmh_native as Nn
    K = N = 128
    x = torch.eye(K)
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 1000.0
    out = Nn.linear_gemm(x.to(_dev()), w.to(_dev()))
    torch.testing.assert_close(out.cpu(), w.t().contiguous(), rtol=0, atol=0)
    xb, wb = x.bfloat16(), (torch.arange(N * K) % 251 - 125).float().reshape(N, K).bfloat16()
    out = Nn.linear_gemm(xb.to(_dev()), wb.to(_dev()))
    torch.testing.assert_close(out.cpu(), wb.float().t().contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("M,d", [(5, 128), (300, 512), (1000, 768), (7, 1024)])
def test_layernorm(M, d):
    import cmh_native as Nn
    g = torch.Generator().manual_seed(d)
    x = torch.randn(M, d, generator=g) * 3 + 0.5
    w, b = torch.randn(d, generator=g), torch.randn(d, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (d,), w.double(), b.double(), 1e-5)
    out = Nn.layernorm(x.to(_dev()), w.to(_dev()), b.to(_dev()))
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    outb = Nn.layernorm(x.to(_dev()), w.to(_dev()), b.to(_dev()), out_bf16=True)
    torch.testing.assert_close(outb.cpu().double(), ref, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("B,T,d,causal", [(2, 5, 128, 0), (3, 50, 768, 0), (2, 77, 512, 1), (2, 16, 128, 1),
                                          (1, 130, 128, 1), (2, 9, 128, 1)])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_attention(B, T, d, causal, mode):
    import cmh_native as Nn
    g = torch.Generator().manual_seed(T * 3 + d)
    qkv = torch.randn(B * T, 3 * d, generator=g)
    kpm = None
    if causal and T >= 9:
        kpm = torch.zeros(B, T, dtype=torch.bool)
        kpm[0, T - 3:] = True                      # MITH-style padding mask on the tail
    src = qkv.bfloat16() if mode == "bf16" else qkv
    h = d // 64
    q, k, v = (src.double()[:, i * d:(i + 1) * d].reshape(B, T, h, 64).permute(0, 2, 1, 3) for i in range(3))
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((T, T), float("-inf"), dtype=torch.float64).triu(1)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    out = Nn.attention(src.to(_dev()), B, T, causal, None if kpm is None else kpm.to(_dev()))
    tol = dict(rtol=1e-2, atol=1e-2) if mode == "bf16" else dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out.cpu().double(), ref, **tol)


@pytest.mark.parametrize("M,N,K", [(160, 256, 64), (333, 512, 512), (2000, 768, 768), (700, 768, 3072), (1300, 512, 2048)])
def test_gemm_fp16_residual_stream(M, N, K):
    """bf16 mode's residual GEMM: out_f16 = fp16(x @ w.T + bias + fp16 residual), in place as the towers run it
    (CMH_EPI_RES_F16 | CMH_EPI_OUT_F16); covers the residual pre-load (short K), the epilogue add (long K), partial tiles
    and workgroups with more than one tile."""
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M + 3 * N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g)
    r = (3.0 * torch.randn(M, N, generator=g)).half()
    ref = x.double() @ w.double().t() + b.double() + r.double()
    out = Nn.linear_gemm(x.to(_dev()), w.to(_dev()), bias=b.to(_dev()), residual=r.to(_dev()), out_f16=True)
    assert out.dtype == torch.float16
    # fp16 output: half an ulp of |ref| (<= 2^-11 relative) on top of the bf16-operand accumulation noise
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1.5e-3, atol=2e-3)
    # f32 residual + fp16 out and fp16 residual + f32 out are legal combinations too
    out2 = Nn.linear_gemm(x.to(_dev()), w.to(_dev()), bias=b.to(_dev()), residual=r.float().to(_dev()), out_f16=True)
    torch.testing.assert_close(out2.cpu().double(), ref, rtol=1.5e-3, atol=2e-3)
    out3 = Nn.linear_gemm(x.to(_dev()), w.to(_dev()), bias=b.to(_dev()), residual=r.to(_dev()))
    assert out3.dtype == torch.float32
    torch.testing.assert_close(out3.cpu().double(), ref, rtol=1e-3, atol=2e-3)
    with pytest.raises(Nn.NativeError):
        Nn.linear_gemm(x.to(_dev()), w[:128].to(_dev()), residual=r[:, :128].contiguous().to(_dev()), out_f16=True)


