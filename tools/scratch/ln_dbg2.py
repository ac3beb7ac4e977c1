import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "clip-based-cross-modal-hashing_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cmh_native as N
from lnfold_bench import timeit, DEV
g = torch.Generator().manual_seed(0)
M, d, No = 12800, 768, 2304
w = (torch.randn(No, d, generator=g) * d ** -0.5).to(DEV)
b = torch.randn(No, generator=g).to(DEV)
gamma = torch.ones(d).to(DEV); beta = torch.zeros(d).to(DEV)
wf, bf, cf = N.ln_fold_weight(w, gamma, beta, b)
wb = w.bfloat16()
part = torch.zeros(3, M, 2, device=DEV); part[:, :, 1] = 256.0
xr = torch.randn(M, d, generator=g)
for tag, x in (("randn", xr), ("zeros", torch.zeros(M, d)), ("3*randn+1", 3 * xr + 1), ("small 0.01", 0.01 * xr)):
    xh = x.half().to(DEV); xb = x.bfloat16().to(DEV)
    t0 = timeit(lambda: N.linear_gemm(xb, wb, bias=b, out_bf16=True))
    t1 = timeit(lambda: N.linear_gemm_ln_consumer(xh, part, wf, bf, cf))
    # plain bf16 kernel fed the fp16 BITS (same toggling as the fold sees, bf16 instruction)
    t2 = timeit(lambda: N.linear_gemm(xh.view(torch.bfloat16), wf.view(torch.bfloat16), bias=b, out_bf16=True))
    print(f"{tag:12s} plain bf16 {t0:7.2f}   fold f16 {t1:7.2f}   plain kernel on the fp16 bits {t2:7.2f}")
