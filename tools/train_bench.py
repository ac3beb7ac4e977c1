"""DSPH training step at BASELINE configs[1] (ViT-B/32, batch 256, 77 tokens, 64 bits): tape forward -> HyP loss -> backward
through heads and both towers -> fused BertAdam (+ SGD on the proxies).  Prints ms per step and a phase breakdown."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import recipe
from model.base.model import CLIP
from model.base.optimization import BertAdam
from model.modelbase import LinearHash
from streams import overlapped
from train.DSPH.loss import HyP

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--forward-only", action="store_true", help="tape forward + loss only (for profiling)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
clip = CLIP(**recipe.CLIP_VITB32).to(dev).float().set_gemm_dtype(a.dtype)
hi, ht = LinearHash(512, 64).to(dev), LinearHash(512, 64).to(dev)
hyp = HyP(numclass=24, output_dim=64, hypseed=0, alpha=0.8).to(dev)
params = [p for n, p in clip.named_parameters() if n != "logit_scale"]
opt = BertAdam([{"params": params, "lr": 1e-5}, {"params": list(hi.parameters()) + list(ht.parameters()), "lr": 1e-3}], lr=1e-3,
               warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=1000, weight_decay=0.2, max_grad_norm=1.0)
sgd = torch.optim.SGD(hyp.parameters(), lr=0.02, momentum=0.9, weight_decay=0.0005)
B = a.batch
img = torch.randn(B, 3, 224, 224, device=dev)
txt = torch.from_numpy(recipe.captions(B, 77, 49408, 1)).to(dev)
lab = (torch.rand(B, 24, device=dev) < 0.15).float()


def step(timers=None):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    fi, ft = overlapped(lambda: clip.encode_image(img), lambda: clip.encode_text(txt))
    loss = hyp(hi(fi), ht(ft), lab)
    ev[1].record()
    if a.forward_only:
        torch.cuda.synchronize()
        if timers is not None:
            timers[0] += ev[0].elapsed_time(ev[1])
        return float(loss)
    opt.zero_grad(); sgd.zero_grad()
    loss.backward()
    ev[2].record()
    opt.step(); sgd.step()
    ev[3].record()
    torch.cuda.synchronize()
    if timers is not None:
        for k in range(3):
            timers[k] += ev[k].elapsed_time(ev[k + 1])
    return float(loss)


for _ in range(2):
    l0 = step()
tm = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(a.steps):
    l1 = step(tm)
dt = (time.perf_counter() - t0) / a.steps
print(f"train step: {dt * 1e3:.2f} ms  ({B / dt:.0f} pairs/s)  forward+loss {tm[0] / a.steps:.2f} ms, backward {tm[1] / a.steps:.2f} ms, "
      f"optimizer {tm[2] / a.steps:.2f} ms; loss {l0:.4f} -> {l1:.4f}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
