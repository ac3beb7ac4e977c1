#!/usr/bin/env python3
"""HyP loss forward time vs (global) batch: what the loss leg of an N-GPU step costs after the all-gather."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch
from train.DSPH.loss import HyP
dev = "cuda:0"
hyp = HyP(numclass=24, output_dim=64, hypseed=0, alpha=0.8).to(dev)
for B in (256, 512, 1024, 2048):
    x, y = torch.tanh(torch.randn(B, 64, device=dev)), torch.tanh(torch.randn(B, 64, device=dev))
    lab = (torch.rand(B, 24, device=dev) < 0.15).float()
    with torch.no_grad():
        for _ in range(3):
            hyp(x, y, lab)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hyp(x, y, lab)
        e1.record()
        torch.cuda.synchronize()
    print(f"B={B:5d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
