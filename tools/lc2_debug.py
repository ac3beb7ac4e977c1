"""lc2 (cmh_set_gemm_lc(4)) against the wide kernel on a few shapes, with the positions of any mismatch: python tools/lc2_debug.py"""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
for (M, Nn, K, res) in ((300, 256, 1088, 1), (12800, 768, 3072, 1), (300, 256, 1088, 0), (12800, 2304, 768, 0), (12800, 768, 768, 1)):
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV); w = (torch.randn(Nn, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
    b = torch.randn(Nn, generator=g).to(DEV); r = torch.randn(M, Nn, generator=g).half().to(DEV) if res else None
    N.set_gemm_rows(0)
    kw = dict(bias=b, residual=r, out_f16=True) if res else dict(bias=b, out_bf16=True)
    N.set_gemm_lc(0); ref = N.linear_gemm(x, w, **kw)
    N.set_gemm_lc(4); got = N.linear_gemm(x, w, **kw)
    N.set_gemm_lc(0)
    bad = (ref != got) | (got != got)
    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
    print((M, Nn, K, res), "bad", int(bad.sum()), "rows", rows[:12].tolist(), "n rows", len(rows), "n cols", len(cols))
