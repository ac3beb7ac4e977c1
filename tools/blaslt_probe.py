"""Reference point only (not used by the product): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the
encoder's shapes, bf16, no epilogue.  Tells how much headroom the hand-written wide kernel still has."""
import torch
dev = torch.device("cuda:0")
shapes = {"v_patch": (12544, 768, 3072), "v_qkv": (12800, 2304, 768), "v_out": (12800, 768, 768), "v_fc1": (12800, 3072, 768),
          "v_fc2": (12800, 768, 3072), "t_qkv": (19712, 1536, 512), "t_out": (19712, 512, 512), "t_fc1": (19712, 2048, 512),
          "t_fc2": (19712, 512, 2048), "sq4096": (4096, 4096, 4096), "sq8192": (8192, 8192, 8192)}
tot_f = tot_t = 0.0
for name, (M, N, K) in shapes.items():
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    for _ in range(30): y = x @ w.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): y = x @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d} {us:9.2f} us {2.0*M*N*K/us/1e6:8.1f} TF/s", flush=True)
    if not name.startswith("sq"): tot_f += 2.0 * M * N * K; tot_t += us
print(f"encoder-shape mix: {tot_f/tot_t/1e6:.1f} TF/s")
