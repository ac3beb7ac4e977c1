"""Where the HOST time of a small-batch training step goes: cProfile of the DCHMT trainer's own train_epoch (configs[0], batch 32)
next to the GPU's busy time for the same steps (torch profiler kernel sum).  usage (GPU box): python tools/step_host_profile.py [--batch 32]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench                      # noqa: E402  (puts the package and the test helpers on sys.path)
import bench_configs              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--pairs", type=int, default=512)
ap.add_argument("--top", type=int, default=35)
a = ap.parse_args()
dev = torch.device("cuda:0")
tr = bench_configs.dchmt_trainer(dev, a.pairs, a.batch, 16)
batch0 = [t for t in next(iter(tr.train_loader))][:3]
for _ in range(3):
    tr._step(*batch0)
torch.cuda.synchronize()
n = len(tr.train_loader)
t0 = time.perf_counter(); tr.train_epoch(0); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"train_epoch: {n} steps, {(t1 - t0) / n * 1e3:.2f} ms per step (batch {a.batch})")
# the same step without the loader and the logger
t0 = time.perf_counter()
for _ in range(n):
    tr._step(*batch0)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"_step on one resident batch: {(t1 - t0) / n * 1e3:.2f} ms per step")
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    for _ in range(8):
        tr._step(*batch0)
    torch.cuda.synchronize()
ev = prof.key_averages()
gpu_us = sum(getattr(e, "self_device_time_total", 0) for e in ev)
print(f"GPU busy: {gpu_us / 8 / 1e3:.2f} ms per step (kernel + copy time summed over 8 steps)")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    tr._step(*batch0)
torch.cuda.synchronize()
pr.disable()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).strip_dirs().sort_stats("tottime").print_stats(a.top)
print("\n".join(l[:170] for l in sio.getvalue().splitlines()))
