"""Attention micro-benchmark at the encoder's shapes (cmh_attention, bf16): us per call and effective HBM rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
dev = torch.device("cuda:0")
for name, (B, T, d, causal) in {"vision": (256, 50, 768, 0), "text": (256, 77, 512, 1)}.items():
    qkv = torch.randn(B * T, 3 * d, device=dev).bfloat16()
    for _ in range(20): N.attention(qkv, B, T, causal)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): N.attention(qkv, B, T, causal)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    mb = B * T * d * 2 * 4 / 1e6
    print(f"{name:7s} B={B} T={T} d={d}: {us:7.2f} us  {mb / us * 1e6 / 1e6:6.2f} TB/s ({mb:.0f} MB)", flush=True)
