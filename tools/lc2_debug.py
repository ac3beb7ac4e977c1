"""lc2 (cmh_set_gemm_lc(4)) against the wide kernel on a few shapes, with the positions of any mismatch: python tools/lc2_debug.py [zero]"""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
zero = len(sys.argv) > 1
for (M, Nn, K, res) in ((300, 256, 1088, 1), (256, 256, 1088, 1), (12800, 768, 3072, 1)):
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV); w = (torch.randn(Nn, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
    b = torch.randn(Nn, generator=g).to(DEV); r = (torch.zeros(M, Nn) if zero else torch.randn(M, Nn, generator=g)).half().to(DEV)
    N.set_gemm_rows(0)
    for rep in range(3):
        N.set_gemm_lc(0); ref = N.linear_gemm(x, w, bias=b, residual=r, out_f16=True)
        N.set_gemm_lc(4); got = N.linear_gemm(x, w, bias=b, residual=r, out_f16=True)
        N.set_gemm_lc(0)
        bad = (ref != got) | (got != got)
        rows = bad.any(1).nonzero().flatten()
        print((M, Nn, K), "rep", rep, "bad", int(bad.sum()), "rows", rows[:10].tolist(), "n rows", len(rows))
