import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "clip-based-cross-modal-hashing_amd"))
import torch
import cmh_native as N
DEV = torch.device("cuda:0")
for (M, d) in [(160, 256), (333, 512)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, d, generator=g).bfloat16()
    wo = (torch.randn(d, d, generator=g) * d ** -0.5).bfloat16()
    bo = torch.randn(d, generator=g)
    r = (2.0 * torch.randn(M, d, generator=g) + 0.7).half()
    x16, part = N.linear_gemm_ln_producer(a.to(DEV), wo.to(DEV), bo.to(DEV), r.to(DEV))
    xd = x16.cpu().double()
    s1 = xd.view(M, d // 256, 256).sum(2).t()
    s2 = (xd * xd).view(M, d // 256, 256).sum(2).t()
    p = part.cpu()
    print("M,d", M, d)
    for m in (0, 1, 15, 16, 17, 79, 80, 81, 159):
        print(m, "got", p[0, m].tolist(), "want", s1[0, m].item(), s2[0, m].item())
    # per-64-column chunk sums for row 0 and row 1
    for m in (0, 1):
        print("row", m, "chunks", xd[m, :256].view(4, 64).sum(1).tolist())
        print("row", m, "16-col", xd[m, :64].view(4, 16).sum(1).tolist())
