#!/usr/bin/env python3
"""Image-preprocessing throughput: cmh_image_preprocess on a batch of MIRFlickr-sized images (500x375 / 375x500 mix) next to
the reference's CPU chain (Pillow resize + crop + float normalise, one process) on a sample of the same images.
   python tools/prep_bench.py [--batch 256] [--iters 20]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dataset.gpu_transform import RaggedImages, preprocess  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cpu-sample", type=int, default=32)
a = ap.parse_args()
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, ((375, 500) if i % 3 else (500, 375)) + (3,), dtype=np.uint8) for i in range(a.batch)]
host = RaggedImages.from_arrays(imgs)
dev = host.to("cuda:0")
for train in (True, False):
    for _ in range(3):
        out = preprocess(dev, 224, train)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        out = preprocess(dev, 224, train)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    by = host.pixels.numel() + out.numel() * 4
    t0 = time.time()
    for _ in range(5):
        d2 = host.to("cuda:0")
        out = preprocess(d2, 224, train)
    torch.cuda.synchronize()
    ms_h2d = (time.time() - t0) / 5 * 1e3
    print(f"{'train' if train else 'eval '} chain: {ms:7.3f} ms / {a.batch} images = {a.batch / ms * 1e3:9.0f} images/s resident "
          f"({by / ms / 1e6:6.1f} GB/s in+out); with the pinned H2D copy of the raw pixels {ms_h2d:7.3f} ms = {a.batch / ms_h2d * 1e3:9.0f} images/s")
try:
    from PIL import Image
    torch.set_num_threads(1)                      # one DataLoader worker = one core
    t0 = time.time()
    for k, arr in enumerate(imgs[:a.cpu_sample + 4]):
        if k == 4:
            t0 = time.time()                          # the first images pay for imports and allocator warm-up
        im = Image.fromarray(arr)
        w, h = im.size
        nw, nh = (224, int(224 * h / w)) if w <= h else (int(224 * w / h), 224)
        im = im.resize((nw, nh), Image.BICUBIC)
        top, left = int(round((nh - 224) / 2.0)), int(round((nw - 224) / 2.0))
        t = torch.from_numpy(np.asarray(im.crop((left, top, left + 224, top + 224))).copy()).permute(2, 0, 1).float().div(255)
        t.sub_(torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(-1, 1, 1)).div_(torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(-1, 1, 1))
    dt = time.time() - t0
    print(f"CPU chain (Pillow {Image.__version__ if hasattr(Image, '__version__') else ''} + torch, 1 process): {a.cpu_sample / dt:7.1f} images/s")
except ImportError:
    print("Pillow not importable: no CPU figure")
