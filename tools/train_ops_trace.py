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
# where the fills come from: Python stacks of aten::zero_ / aten::fill_ / aten::zeros (grouped by the innermost repo frame)
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof2:
    g["step"]()
import collections
where = collections.Counter()
for ev in prof2.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like"):
        frames = [f for f in (ev.stack or []) if "site-packages/torch" not in f and "<built-in" not in f]
        where[(ev.name, frames[0] if frames else "?")] += 1
for (name, fr), c in where.most_common(15):
    print(f"{c:5d} x {name:18s} {fr}")
