"""Which ATen operators still run inside a DSPH training step (everything else is libcmh kernels): torch.profiler table of one step.
   python tools/train_ops_trace.py"""
import os, sys, runpy
sys.argv = ["train_bench.py", "--steps", "2"]
ROOT = os.path.dirname(os.path.abspath(__file__))
g = runpy.run_path(os.path.join(ROOT, "train_bench.py"))
import torch
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    g["step"]()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
for e in rows[:30]:
    print(f"{e.key:40s} calls {e.count:5d}  device {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)):9.1f} us  cpu {e.cpu_time_total:9.1f} us")
