#!/usr/bin/env python3
"""GEMM micro-benchmark over the encoder's shapes (through the C ABI, cmh_linear_gemm).
   python tools/gemm_bench.py [--dtype bf16|f32] [--iters 20] [--shapes vision|text|all]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch  # noqa: E402

import cmh_native as N  # noqa: E402

SHAPES = {
    "v_patch": (12544, 768, 3072), "v_qkv": (12800, 2304, 768), "v_out": (12800, 768, 768),
    "v_fc1": (12800, 3072, 768), "v_fc2": (12800, 768, 3072),
    "t_qkv": (19712, 1536, 512), "t_out": (19712, 512, 512), "t_fc1": (19712, 2048, 512), "t_fc2": (19712, 512, 2048),
    "sq4096": (4096, 4096, 4096),
}

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--only", default="")
ap.add_argument("--quickgelu", action="store_true", help="bias + QuickGELU epilogue (the c_fc launches)")
a = ap.parse_args()
dev = torch.device("cuda:0")
tot_t = tot_f = 0.0
for name, (M, Nn, K) in SHAPES.items():
    if a.only and name not in a.only.split(","):
        continue
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(Nn, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(Nn, device=dev)
    for _ in range(max(3, a.iters // 4)):   # also lets the clocks settle
        N.linear_gemm(x, w, bias=b, quickgelu=a.quickgelu, out_bf16=(a.dtype == "bf16"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        N.linear_gemm(x, w, bias=b, quickgelu=a.quickgelu, out_bf16=(a.dtype == "bf16"))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    fl = 2.0 * M * Nn * K
    if name != "sq4096":
        tot_t += us
        tot_f += fl
    print(f"{name:8s} M={M:6d} N={Nn:5d} K={K:5d}  {us:8.2f} us  {fl / us / 1e6:8.1f} TF/s", flush=True)
if tot_t:
    print(f"encoder-shape mix: {tot_f / tot_t / 1e6:.1f} TF/s")
