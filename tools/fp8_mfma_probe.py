"""How does v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, block scales 2^0) add its 128 products?  One output element gets one
big product (448 * 448 = 200 704) and 127 equal small ones (2^-s * 2^-s, exactly representable in e4m3 for s <= 6, all exact in f32);
an f32 accumulation chain would return 200704 + 127 * 4^-s exactly whenever that sum fits 24 bits.  Through the C ABI
(cmh_linear_gemm_fp8, K = 128: exactly one scaled MFMA per output element)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
dev = torch.device("cuda:0")
M, Nn, K = 160, 256, 128
print("small product   exact sum (f64)        f32(exact)          MFMA result         lost")
for s in range(0, 7):
    x = torch.zeros(M, K); w = torch.zeros(Nn, K)
    x[0, 0] = 448.0; w[0, 0] = 448.0
    x[0, 1:] = 2.0 ** -s; w[0, 1:] = 2.0 ** -s
    x8 = N.fp8_quantize(x.to(dev), 1.0)                       # scale 1: the values are e4m3 numbers already
    w8, cs = N.fp8_quantize_weight(w.to(dev))                 # per-row scale = amax / 448 = 1 for row 0
    out = N.linear_gemm_fp8(x8, w8, torch.ones(Nn, device=dev), 1.0)
    exact = 448.0 * 448.0 + 127 * 4.0 ** -s
    got = float(out[0, 0])
    print(f"  4^-{s}          {exact:22.10f} {float(torch.tensor(exact, dtype=torch.float32)):18.6f} {got:18.6f} {exact - got:12.6f}")
# random operands: error of one MFMA against the f64 sum of the same quantised operands, in units of the largest product
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g); w = torch.randn(Nn, K, generator=g)
x8 = N.fp8_quantize(x.to(dev), 4.0 / 448); w8, cs = N.fp8_quantize_weight(w.to(dev))
xq = N.fp8_dequantize(x8, 4.0 / 448).double().cpu(); wq = (N.fp8_dequantize(w8, 1.0).double().cpu() * cs.double().cpu()[:, None])
ref = xq @ wq.t()
out = N.linear_gemm_fp8(x8, w8, cs, 4.0 / 448).double().cpu()
err = (out - ref).abs()
big = (xq.abs().max(1).values[:, None] * wq.abs().max(1).values[None, :])
print(f"random operands, K = 128: max |err| {float(err.max()):.3e}, max |err| / (row max * col max) {float((err / big).max()):.3e}, "
      f"max rel {float((err / ref.abs().clamp(min=1e-3)).max()):.3e}; an f32 chain of exact products would stay below ~{128 * 2.0 ** -24:.1e} relative")
