import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import preputil as pu
from oracle import preprocess_oracle as po
from dataset.gpu_transform import RaggedImages, preprocess
for (h, w) in ((16, 53), (53, 16), (16, 16)):
    im = pu.image(h, w)
    out, u8 = preprocess(RaggedImages.from_arrays([im]).to("cuda:0"), 16, False, want_u8=True)
    ref = po.transform_u8(im, 16, False)
    got = u8[0].cpu().numpy()
    d = got.astype(int) - ref
    print((h, w), "bad", (d != 0).sum(), "of", d.size)
    if (d != 0).any():
        r, c, ch = [a[0] for a in np.nonzero(d)]
        print(" first bad at", r, c, ch, "got", got[r, c], "ref", ref[r, c])
        print(" row got", got[r, :, ch]); print(" row ref", ref[r, :, ch])
