"""End-to-end training epochs through the real-data path: decoded uint8 images (the reference's npy=True layout) -> DataLoader ->
native batch tokenizer + ragged batch -> GPU image transform -> DSPH step (ViT-B/32, bf16, batch 256).  Epoch 1 runs the
transform, later epochs are served from the device-resident cache of resized images.  Needs the CLIP merges file for real
captions; falls back to the miniature merges of the tests."""
import argparse, gzip, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, recipe
import bpeutil as bu
from dataset.base import BaseDataset, DeviceLoader
from model.base.model import CLIP
from model.base.optimization import BertAdam
from model.base.simple_tokenizer import default_bpe
from model.modelbase import LinearHash
from streams import overlapped
from train.DSPH.loss import HyP

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=2048)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--workers", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
vocab = default_bpe()
if not os.path.exists(vocab):
    vocab = os.path.join(tempfile.mkdtemp(), "mini.txt.gz")
    with gzip.open(vocab, "wb") as f:
        f.write(open(bu.MINI_MERGES, "rb").read())
rng = np.random.default_rng(0)
images = np.empty(a.items, dtype=object)
for i in range(a.items):
    images[i] = rng.integers(0, 256, ((375, 500) if i % 3 else (500, 375)) + (3,), dtype=np.uint8)
words = bu.CORPUS.split()
captions = np.array([[" ".join(rng.choice(words, size=int(rng.integers(5, 20))))] for _ in range(a.items)])
labels = (rng.random((a.items, 24)) < 0.15).astype(np.float32)
data = BaseDataset(captions, images, labels, is_train=True, maxWords=32, imageResolution=224, npy=True, bpe_path=vocab)
loader = DeviceLoader(data, dev, batch_size=a.batch, shuffle=True, num_workers=a.workers, drop_last=True)
torch.manual_seed(0)
clip = CLIP(**recipe.CLIP_VITB32).to(dev).float().set_gemm_dtype("bf16")
# the miniature vocabulary has < 1000 ids; the full one needs the full embedding table (49408 rows), which CLIP_VITB32 has
hi, ht = LinearHash(512, 64).to(dev), LinearHash(512, 64).to(dev)
hyp = HyP(numclass=24, output_dim=64, hypseed=0, alpha=0.8).to(dev)
params = [p for n, p in clip.named_parameters() if n != "logit_scale"]
opt = BertAdam([{"params": params, "lr": 1e-5}, {"params": list(hi.parameters()) + list(ht.parameters()), "lr": 1e-3}], lr=1e-3, warmup=0.1,
               schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=1000, weight_decay=0.2, max_grad_norm=1.0)
sgd = torch.optim.SGD(hyp.parameters(), lr=0.02, momentum=0.9, weight_decay=0.0005)
for epoch in range(a.epochs):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for image, text, label, index in loader:
        text, label = text.to(dev, non_blocking=True), label.to(dev, non_blocking=True)
        fi, ft = overlapped(lambda: clip.encode_image(image), lambda: clip.encode_text(text))
        loss = hyp(hi(fi), ht(ft), label)
        opt.zero_grad(); sgd.zero_grad()
        loss.backward()
        opt.step(); sgd.step()
        n += image.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"epoch {epoch}: {n} pairs in {dt:.2f} s = {n / dt:.0f} pairs/s  ({'cached images' if loader.cached_epochs and epoch >= 1 else 'decode-free npy images, GPU transform'}); loss {float(loss.detach()):.4f}", flush=True)
