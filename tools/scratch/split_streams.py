"""How many streams should a step's towers be cut into?  (image tower in 1 / 2 parts, text tower in 1 / 2 parts, one stream each)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch
import cmh_native as N
from bench import VITB32, synthetic_batch
from model.base.model import CLIP
from model.modelbase import LinearHash
from streams import overlapped

dev = torch.device("cuda:0")
torch.manual_seed(1814)
clip = CLIP(**VITB32).to(dev).float().set_gemm_dtype("bf16")
clip.assume_frozen = True
ih, th = LinearHash(512, 64).to(dev).eval(), LinearHash(512, 64).to(dev).eval()
image, text, label = synthetic_batch(256, 77, 24, 1814, dev)


def tower(enc, head, x):
    h = head(enc(x))
    N.pack_codes(N.sign_codes(h), validate=False)
    return h


def step(ni, nt):
    with torch.no_grad():
        fns = [(lambda x=x: tower(clip.encode_image, ih, x)) for x in image.chunk(ni)] + \
              [(lambda x=x: tower(clip.encode_text, th, x)) for x in text.chunk(nt)]
        return overlapped(*fns)


ref = None
for ni, nt in [(1, 1), (2, 1), (2, 2), (1, 2), (4, 2), (1, 1), (2, 1)]:
    for _ in range(6):
        out = step(ni, nt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        out = step(ni, nt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    hi = torch.cat(out[:ni]); ht = torch.cat(out[ni:])
    if ref is None:
        ref = (hi.clone(), ht.clone())
    same = torch.equal(hi, ref[0]) and torch.equal(ht, ref[1])
    print(f"image x{ni} text x{nt}: {dt * 1e3:7.3f} ms/step  {256 / dt:9.0f} pairs/s   identical codes: {same}", flush=True)
