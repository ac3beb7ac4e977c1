import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
dev = torch.device("cuda:0")
for name, (M, Nn, K) in {"v_qkv": (12800, 2304, 768), "v_fc1": (12800, 3072, 768), "t_fc1": (19712, 2048, 512), "v_fc2": (12800, 768, 3072)}.items():
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(Nn, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(Nn, device=dev); out = torch.empty(M, Nn, dtype=torch.bfloat16, device=dev)
    for epi in [int(e) for e in os.environ.get("EPIS", "9,521,265,9").split(",")]:
        def run():
            N.check(N.lib().cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), N.ptr(b), None, N.ptr(out), M, Nn, K, epi, N.stream_ptr(dev)), "gemm")
        for _ in range(30): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(100): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 100
        print(f"{name} epi={epi:3d} {us:8.2f} us {2.0*M*Nn*K/us/1e6:8.1f} TF/s", flush=True)
