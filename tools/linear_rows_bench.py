"""cmh_linear_act on MITH's token-level shapes: us per call, many-row kernel against the one-row-per-block kernel (slices)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "clip-based-cross-modal-hashing_amd"))
import torch
import cmh_native as N
dev = "cuda:0"
for M, Nn, K in ((12544, 64, 512), (8192, 64, 512), (19712, 64, 512)):
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) * K ** -0.5
    def t(f, n=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    full = t(lambda: N.linear_act(x, w, None, N.ACT_TANH))
    xs = [x[i:i + 2000].contiguous() for i in range(0, M, 2000)]
    parts = t(lambda: [N.linear_act(c, w, None, N.ACT_TANH) for c in xs])
    print(f"{M} x {K} -> {Nn}: many-row kernel {full:.1f} us, one row per block (in slices of 2000 rows) {parts:.1f} us")
