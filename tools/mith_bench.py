"""MITH forward at batch 128 (ViT-B/32 trunk with all tokens + HashingModel): ms per forward and the split trunk / hashing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, recipe
from types import SimpleNamespace
from model.MITH import HashingModel, build_model
dev = torch.device("cuda:0")
torch.manual_seed(0)
sd = {k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(recipe.CLIP_VITB32, 1).items()}
clip = build_model(sd).to(dev).float().set_gemm_dtype("bf16")
clip.padded_tokens_unused = True      # as model/MITH.py::MITH sets it (HashingModel is the only reader of the text tokens)
hm = HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=64, dropout=0.0, transformer_layers=2, activation="gelu",
                                                           top_k_label=8, res_mlp_layers=2)).to(dev).eval()
B = 128
img = torch.randn(B, 3, 224, 224, device=dev)
txt = torch.from_numpy(recipe.captions(B, 32, 49408, 1)).to(dev)
kpm = txt == 0
def fwd():
    with torch.no_grad():
        seq_i, _, cls_i = clip.encode_image(img)
        seq_t, _, nk, eos = clip.encode_text(txt, kpm)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        od = hm(seq_i, seq_t, cls_i, eos, nk)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1, t2
ref = None
for mode in ("f32", "bf16"):
    hm.set_gemm_dtype(mode)
    for _ in range(3): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter(); t1, t2 = fwd()
    print(f"MITH forward B={B}, HashingModel GEMMs {mode}: trunk {(t1 - t0) * 1e3:.2f} ms, HashingModel {(t2 - t1) * 1e3:.2f} ms")
