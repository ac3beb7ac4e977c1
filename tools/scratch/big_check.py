import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "clip-based-cross-modal-hashing_amd"))
import torch
import cmh_native as N
dev = torch.device("cuda:0")
def timeit(fn, rep=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3
for name, (M, Nn, K, act) in {"v_qkv": (12800, 2304, 768, False), "v_fc1": (12800, 3072, 768, True), "t_qkv": (10499, 1536, 512, False),
                              "t_fc1": (10499, 2048, 512, True), "odd": (7000, 2304, 768, True), "longK": (12800, 2304, 3072, False), "longK2": (12800, 3072, 6144, False)}.items():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(Nn, K, generator=g) * K ** -0.5).bfloat16().to(dev)
    b = torch.randn(Nn, generator=g).to(dev)
    N.set_gemm_big(0)
    ref = N.linear_gemm(x, w, bias=b, quickgelu=act, out_bf16=True)
    t0 = timeit(lambda: N.linear_gemm(x, w, bias=b, quickgelu=act, out_bf16=True))
    N.set_gemm_big(1)
    out = N.linear_gemm(x, w, bias=b, quickgelu=act, out_bf16=True)
    t1 = timeit(lambda: N.linear_gemm(x, w, bias=b, quickgelu=act, out_bf16=True))
    same = torch.equal(out, ref)
    nbad = int((out != ref).sum()) if not same else 0
    fl = 2.0 * M * Nn * K
    print(f"{name:6s} M={M} N={Nn} K={K}: wide {t0:7.2f} us ({fl/t0/1e6:7.1f} TF/s)   big {t1:7.2f} us ({fl/t1/1e6:7.1f} TF/s)   identical: {same} ({nbad} differ)", flush=True)
