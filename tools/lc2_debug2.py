import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
M, Nn, K = 256, 256, 1088
x = torch.randn(M, K, generator=g).bfloat16().to(DEV); w = (torch.randn(Nn, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
b = torch.zeros(Nn).to(DEV)
N.set_gemm_rows(0)
for name, r in (("zero residual", torch.zeros(M, Nn).half().to(DEV)), ("residual = 1.0", torch.ones(M, Nn).half().to(DEV)), ("residual = row index / 8", (torch.arange(M)[:, None].float() / 8).expand(M, Nn).contiguous().half().to(DEV))):
    N.set_gemm_lc(0); ref = N.linear_gemm(x, w, bias=b, residual=r, out_f16=True)
    N.set_gemm_lc(4); got = N.linear_gemm(x, w, bias=b, residual=r, out_f16=True)
    N.set_gemm_lc(0)
    bad = (ref != got) | (got != got)
    print(name, "bad", int(bad.sum()))
    d = (got.float() - ref.float())
    for row in (0, 1, 15, 16, 64, 65):
        print("  row", row, "got-ref first 8 cols", [round(v, 3) for v in d[row, :8].tolist()], " cols 32..35", [round(v, 3) for v in d[row, 32:36].tolist()])
