#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from
/root/reference, never copied) on the seeded recipes of tests/golden/recipe.py.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only clip,map,...]

Fixtures store expected OUTPUTS (+ small crafted inputs); weights and bulk inputs are
regenerated from the recipe by the tests.  Pinned to the torch build printed in each
fixture's `meta` (reference arithmetic = PyTorch ATen CPU fp32).

Inert stubs inserted for modules the image lacks (ordinary ModuleNotFoundError, see
SURVEY.md §8c): torch.utils.tensorboard.SummaryWriter, torchvision(.transforms), ftfy,
xlrd (replaced by a 20-line xlsx cell reader so train/DSPH/loss.py:19-20 can read its
threshold table).
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import importlib.util
import os
import re
import sys
import tempfile
import types
import zipfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402

REF = "/root/reference"
sys.dont_write_bytecode = True


# --------------------------------------------------------------------------- stubs
def install_stubs():
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # constructed at model/modelbase.py:52-53, never used
        def __init__(self, *a, **k):
            pass
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize"):
        setattr(tvt, n, lambda *a, **k: None)
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt

    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s
    sys.modules["ftfy"] = ftfy

    xl = types.ModuleType("xlrd")

    class _Cell:
        def __init__(self, v):
            self.value = v

    class _Sheet:
        def __init__(self, path):
            z = zipfile.ZipFile(path)
            xml = z.read("xl/worksheets/sheet1.xml").decode()
            self.cells = {}
            for m in re.finditer(r'<c r="([A-Z]+)(\d+)"[^>]*><v>([^<]*)</v></c>', xml):
                col = 0
                for ch in m.group(1):
                    col = col * 26 + (ord(ch) - 64)
                self.cells[(int(m.group(2)) - 1, col - 1)] = float(m.group(3))

        def row(self, r):
            width = max(c for (rr, c) in self.cells if rr == r) + 1
            return [_Cell(self.cells.get((r, c), "")) for c in range(width)]

    class _Book:
        def __init__(self, path):
            self.path = path

        def sheet_by_index(self, i):
            return _Sheet(self.path)

    xl.open_workbook = lambda p: _Book(p if os.path.isabs(p) else os.path.join(REF, p))
    sys.modules["xlrd"] = xl


def ref_import(name):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return importlib.import_module(name)


def meta():
    return np.array(f"torch {torch.__version__} cpu fp32; numpy {np.__version__}; "
                    f"reference=/root/reference (QinLab-WFU/CLIP-based-Cross-Modal-Hashing)")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, meta=meta(), **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- CLIP trunk
def build_ref_clip(cfg, seed):
    m = ref_import("model.base.model")
    clip = m.CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"],
                  cfg["vision_width"], cfg["vision_patch_size"], cfg["context_length"],
                  cfg["vocab_size"], cfg["transformer_width"], cfg["transformer_heads"],
                  cfg["transformer_layers"])
    sd = {k: t(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}
    clip.load_state_dict(sd, strict=True)
    return clip.float()


def run_clip_with_taps(clip, image, text_list):
    """Forward through the reference CLIP, tapping ln_pre and every resblock."""
    taps = {}
    hooks = []

    def tap(name):
        def fn(mod, inp, out):
            taps.setdefault(name, []).append(out.detach().clone())
        return fn
    hooks.append(clip.visual.ln_pre.register_forward_hook(tap("v_ln_pre")))
    for i, blk in enumerate(clip.visual.transformer.resblocks):
        hooks.append(blk.register_forward_hook(tap(f"v_block{i}")))
    for i, blk in enumerate(clip.transformer.resblocks):
        hooks.append(blk.register_forward_hook(tap(f"t_block{i}")))
    with torch.no_grad():
        img_feat = clip.encode_image(image)
        txt_feats = [clip.encode_text(tx) for tx in text_list]
    for h in hooks:
        h.remove()
    return img_feat, txt_feats, taps


def gen_clip_tiny():
    cfg, seed = recipe.CLIP_TINY, 7
    clip = build_ref_clip(cfg, seed)
    image = t(recipe.images(3, cfg["image_resolution"], seed))
    txt16 = t(recipe.captions(3, 16, cfg["vocab_size"], seed))
    txt9 = t(recipe.captions(3, 9, cfg["vocab_size"], seed + 1))
    img_feat, (tf16, tf9), taps = run_clip_with_taps(clip, image, [txt16, txt9])
    out = dict(seed=np.int64(seed), img_feat=img_feat.numpy(), txt_feat_L16=tf16.numpy(),
               txt_feat_L9=tf9.numpy())
    out["v_ln_pre"] = taps["v_ln_pre"][0].numpy()                       # [B,T,d] (NLD)
    for i in range(cfg["vision_layers"]):
        out[f"v_block{i}"] = taps[f"v_block{i}"][0].permute(1, 0, 2).numpy()   # LND -> NLD
    for i in range(cfg["transformer_layers"]):
        out[f"t_block{i}_L16"] = taps[f"t_block{i}"][0].permute(1, 0, 2).numpy()
        out[f"t_block{i}_L9"] = taps[f"t_block{i}"][1].permute(1, 0, 2).numpy()
    save("clip_tiny.npz", **out)


def gen_clip_vitb32():
    cfg, seed = recipe.CLIP_VITB32, 11
    clip = build_ref_clip(cfg, seed)
    image = t(recipe.images(2, 224, seed))
    txt77 = t(recipe.captions(2, 77, cfg["vocab_size"], seed))
    txt32 = t(recipe.captions(2, 32, cfg["vocab_size"], seed + 1))
    img_feat, (tf77, tf32), taps = run_clip_with_taps(clip, image, [txt77, txt32])
    rows = [0, 1, 25, 49]
    out = dict(seed=np.int64(seed), img_feat=img_feat.numpy(), txt_feat_L77=tf77.numpy(),
               txt_feat_L32=tf32.numpy(), v_rows=np.array(rows))
    out["v_ln_pre_rows"] = taps["v_ln_pre"][0][:, rows].numpy()
    for i in (0, 5, 11):
        out[f"v_block{i}_rows"] = taps[f"v_block{i}"][0].permute(1, 0, 2)[:, rows].numpy()
        out[f"t_block{i}_L77_rows"] = taps[f"t_block{i}"][0].permute(1, 0, 2)[:, rows].numpy()
    save("clip_vitb32.npz", **out)


# --------------------------------------------------------------------------- Baseclip + heads
def _saved_clip_ckpt(cfg, seed):
    sd = {k: t(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}
    f = tempfile.NamedTemporaryFile(suffix=".pt", delete=False)
    f.close()
    torch.save(sd, f.name)
    return f.name


def gen_baseclip_tiny():
    """Reference Baseclip API (model/modelbase.py:38-96) over the tiny CLIP, through
    load_clip -> build_model (fp16 round trip of GEMM weights) -> .float(), eval-mode heads.
    Methods: DSPH (LinearHash), DCHMT (HashLayer), DNPH (LinearHash + Pre_Layer)."""
    cfg, seed = recipe.CLIP_TINY, 7
    ckpt = _saved_clip_ckpt(cfg, seed)
    image = t(recipe.images(3, cfg["image_resolution"], seed))
    text = t(recipe.captions(3, 16, cfg["vocab_size"], seed))
    out = dict(seed=np.int64(seed))
    tmp = tempfile.mkdtemp()
    base = ref_import("train.base")

    for K in (16, 64):
        # ---- DSPH
        MDSPH = ref_import("model.DSPH").MDSPH
        m = MDSPH(outputDim=K, clipPath=ckpt, saveDir=tmp).float()
        for side in ("image", "text"):
            w, b = recipe.head_linear(cfg["embed_dim"], K, seed, f"dsph_{side}_{K}")
            getattr(m, f"{side}_hash").fc.weight.data.copy_(t(w))
            getattr(m, f"{side}_hash").fc.bias.data.copy_(t(b))
        m.eval()
        with torch.no_grad():
            hi, ht = m(image, text)
            fi, ft = m.clip.encode_image(image), m.clip.encode_text(text)
        out[f"dsph_img_K{K}"] = hi.numpy()
        out[f"dsph_txt_K{K}"] = ht.numpy()
        out[f"dsph_img_code_K{K}"] = torch.sign(hi).numpy()
        out[f"dsph_txt_code_K{K}"] = torch.sign(ht).numpy()
        if K == 16:
            out["feat_img_fp16w"] = fi.numpy()
            out["feat_txt_fp16w"] = ft.numpy()

        # ---- DCHMT
        MDCMHT = ref_import("model.DCHMT").MDCMHT
        m = MDCMHT(outputDim=K, clipPath=ckpt, saveDir=tmp).float()
        for side in ("image", "text"):
            hl = getattr(m, f"{side}_hash")
            w, b = recipe.head_linear(cfg["embed_dim"], 128, seed, f"dchmt_{side}_fc_{K}")
            hl.fc.weight.data.copy_(t(w)); hl.fc.bias.data.copy_(t(b))
            w2, b2 = recipe.head_linear(128, 2 * K, seed, f"dchmt_{side}_bits_{K}")
            for j, lin in enumerate(hl.hash_list):
                lin.weight.data.copy_(t(w2[2 * j:2 * j + 2])); lin.bias.data.copy_(t(b2[2 * j:2 * j + 2]))
        m.eval()
        with torch.no_grad():
            li, lt = m(image, text)
        out[f"dchmt_img_K{K}"] = torch.stack(li, 1).numpy()     # [B,K,2]
        out[f"dchmt_txt_K{K}"] = torch.stack(lt, 1).numpy()
        mk = base.TrainBase.make_hash_code_DCHMT
        out[f"dchmt_img_code_K{K}"] = mk(None, li).numpy()
        out[f"dchmt_txt_code_K{K}"] = mk(None, lt).numpy()

    # ---- DNPH (K=16, C=21)
    MDNPH = ref_import("model.DNPH_TOMM").MDNPH
    m = MDNPH(outputDim=16, num_classes=21, clipPath=ckpt, saveDir=tmp).float()
    for side in ("image", "text"):
        w, b = recipe.head_linear(cfg["embed_dim"], 16, seed, f"dnph_{side}_hash")
        getattr(m, f"{side}_hash").fc.weight.data.copy_(t(w)); getattr(m, f"{side}_hash").fc.bias.data.copy_(t(b))
        w, b = recipe.head_linear(cfg["embed_dim"], 21, seed, f"dnph_{side}_pre")
        getattr(m, f"{side}_pre").fc.weight.data.copy_(t(w)); getattr(m, f"{side}_pre").fc.bias.data.copy_(t(b))
    m.eval()
    with torch.no_grad():
        hi, pi, ht, pt = m(image, text)
    out.update(dnph_img=hi.numpy(), dnph_img_pre=pi.numpy(), dnph_txt=ht.numpy(), dnph_txt_pre=pt.numpy())
    os.unlink(ckpt)
    save("baseclip_tiny.npz", **out)


# --------------------------------------------------------------------------- losses
def gen_loss_dsph():
    """train/DSPH/loss.py:10-72 HyP.  get_args() is called without its argument as shipped
    (:13) -> patched to return a Namespace; xlrd -> stub reader (see module docstring)."""
    loss_mod = ref_import("train.DSPH.loss")
    out = {}
    seed = 21
    for (B, K, C, alpha, p) in [(32, 64, 24, 0.8, 0.15), (48, 16, 80, 0.8, 0.05), (16, 128, 21, 0.0, 0.2),
                                (8, 32, 24, 0.8, 0.04)]:
        tag = f"B{B}_K{K}_C{C}"
        loss_mod.get_args = lambda K=K, C=C, alpha=alpha: SimpleNamespace(
            hypseed=0, numclass=C, output_dim=K, alpha=alpha)
        hyp = loss_mod.HyP()
        out[f"{tag}_threshold"] = np.float64(hyp.threshold)
        out[f"{tag}_default_proxies_sha"] = np.array(
            hashlib.sha256(hyp.proxies.detach().numpy().tobytes()).hexdigest())
        out[f"{tag}_default_proxies_head"] = hyp.proxies.detach().numpy()[:2, :8].copy()
        prox = recipe.features(C, K, seed, f"dsph_prox_{tag}")
        hyp.proxies.data.copy_(t(prox))
        x = torch.tanh(t(recipe.features(B, K, seed, f"dsph_x_{tag}"))).requires_grad_()
        y = torch.tanh(t(recipe.features(B, K, seed, f"dsph_y_{tag}"))).requires_grad_()
        lab = t(recipe.labels(B, C, seed, p=p, tag=f"dsph_lab_{tag}"))
        loss = hyp(x, y, lab)
        loss.backward()
        out[f"{tag}_loss"] = loss.detach().numpy()
        out[f"{tag}_gx"] = x.grad.numpy()
        out[f"{tag}_gy"] = y.grad.numpy()
        out[f"{tag}_gprox"] = hyp.proxies.grad.numpy()
        out[f"{tag}_alpha"] = np.float64(alpha)
        out[f"{tag}_p"] = np.float64(p)
    save("loss_dsph.npz", **out)


def gen_loss_dchmt():
    """train/DCHMT/hash_train.py:82-150 similarity_loss/our_loss driven as unbound methods
    on a SimpleNamespace self (trainers cannot be constructed on CPU)."""
    tr = ref_import("train.DCHMT.hash_train").DCHMTTrainer
    out = {}
    seed = 31
    for (B, K, C, fn, lt) in [(32, 16, 24, "euclidean", "l2"), (32, 16, 24, "cosine", "l2"),
                              (24, 64, 24, "euclidean", "l1"), (24, 64, 80, "cosine", "l1")]:
        tag = f"B{B}_K{K}_C{C}_{fn}_{lt}"
        args = SimpleNamespace(vartheta=0.5, sim_threshold=0.1, similarity_function=fn, loss_type=lt,
                               output_dim=K, hash_layer="select", display_step=50, epochs=1)
        self = SimpleNamespace(args=args, rank="cpu", global_step=1, logger=None)
        self.similarity_loss = lambda a, b, s, self=self: tr.similarity_loss(self, a, b, s)
        # softmax-pair outputs like HashLayer produces (model/DCHMT.py:20-26), cat to [B,2K]
        zi = t(recipe.features(B, 2 * K, seed, f"dchmt_zi_{tag}")).view(B, K, 2)
        zt = t(recipe.features(B, 2 * K, seed, f"dchmt_zt_{tag}")).view(B, K, 2)
        hi = torch.softmax(2 * zi, -1).reshape(B, 2 * K).requires_grad_()
        ht = torch.softmax(2 * zt, -1).reshape(B, 2 * K).requires_grad_()
        lab = t(recipe.labels(B, C, seed, tag=f"dchmt_lab_{tag}"))
        loss = tr.our_loss(self, hi, ht, lab, 0, 1)
        loss.backward()
        out[f"{tag}_loss"] = loss.detach().numpy()
        out[f"{tag}_gi"] = hi.grad.numpy()
        out[f"{tag}_gt"] = ht.grad.numpy()
    save("loss_dchmt.npz", **out)


# --------------------------------------------------------------------------- mAP
def gen_map():
    cu = ref_import("utils.calc_utils")
    out = {}
    cases = [  # name, Q, N, K, C, k, kind, zeros
        ("rand_1k_16", 1000, 1000, 16, 24, None, "rand", 0),
        ("rand_1k_64", 1000, 1000, 64, 24, None, "rand", 0),
        ("corr_1k_64", 500, 1000, 64, 24, None, "corr", 0),
        ("corr_1k_64_k50", 500, 1000, 64, 24, 50, "corr", 0),
        ("zeros_300_32", 100, 300, 32, 10, None, "rand", 40),
        ("tiny_17_16", 20, 17, 16, 5, None, "rand", 0),
        ("tiny_16_8", 20, 16, 8, 5, None, "rand", 0),
        ("odd_5003_128", 64, 5003, 128, 21, None, "corr", 0),
        ("flickr_20015_64", 200, 20015, 64, 24, None, "corr", 0),
        ("nus_190k_128", 24, 190000, 128, 21, None, "corr", 0),
        ("coco_117k_64", 16, 117218, 64, 80, 1000, "corr", 0),
    ]
    for name, Q, N, K, C, k, kind, zeros in cases:
        seed = 1234
        p = 0.15 if C < 50 else 0.04
        qL = recipe.labels(Q, C, seed, p=p, tag=f"map_qL_{name}")
        rL = recipe.labels(N, C, seed, p=p, tag=f"map_rL_{name}")
        if kind == "rand":
            qB = recipe.sign_codes(Q, K, seed, f"map_qB_{name}", zeros=zeros)
            rB = recipe.sign_codes(N, K, seed, f"map_rB_{name}", zeros=zeros * 3)
        else:
            qB = recipe.correlated_codes(qL, K, seed, f"map_qB_{name}")
            rB = recipe.correlated_codes(rL, K, seed, f"map_rB_{name}")
        qBt, rBt, qLt, rLt = t(qB), t(rB), t(qL), t(rL)
        mAP = cu.calc_map_k_matrix(qBt, rBt, qLt, rLt, k)
        ap = np.zeros(Q, np.float32)
        for i in range(Q):   # per-query AP = the reference function on a 1-query problem
            ap[i] = float(cu.calc_map_k_matrix(qBt[i:i + 1], rBt, qLt[i:i + 1], rLt, k))
        out[f"{name}_shape"] = np.array([Q, N, K, C, -1 if k is None else k], np.int64)
        out[f"{name}_kind"] = np.array(kind)
        out[f"{name}_zeros"] = np.int64(zeros)
        out[f"{name}_p"] = np.float64(p)
        out[f"{name}_map"] = np.float32(mAP)
        out[f"{name}_ap"] = ap
        # permutations (utils/calc_utils.py:30-31) for a few queries
        nperm = 8 if N <= 5003 else 3
        sha = []
        for i in range(nperm):
            hamm = cu.calc_hammingDist(qBt[i, :], rBt)
            _, ind = torch.sort(hamm)
            ind = ind.squeeze().numpy()
            if N <= 5003:
                out[f"{name}_ind{i}"] = ind.astype(np.int32)
            sha.append(hashlib.sha256(ind.astype(np.int64).tobytes()).hexdigest())
        out[f"{name}_ind_sha"] = np.array(sha)
        print(f"  map {name}: mAP={float(mAP):.6f} skipped={(qL @ rL.T > 0).sum(1).min() == 0}")
    save("map.npz", **out)


def gen_codetable():
    """DSPH threshold table (train/DSPH/codetable.xlsx, data).  Row n (0-based xlrd row),
    column m: entry read at train/DSPH/loss.py:20 as sheet.row(output_dim)[ceil(log2(numclass))].
    Extracted as numbers into the package's own JSON (columns 0..16)."""
    import json
    import xlrd
    sheet = xlrd.open_workbook("train/DSPH/codetable.xlsx").sheet_by_index(0)
    rows = {}
    for n in range(1, 257):
        if not any(rr == n for (rr, _) in sheet.cells):
            continue
        r = sheet.row(n)
        rows[str(n)] = [None if c.value == "" else c.value for c in r[:17]]
    pkg = os.path.join(os.path.dirname(os.path.dirname(HERE)), "clip-based-cross-modal-hashing_amd",
                       "train", "DSPH")
    os.makedirs(pkg, exist_ok=True)
    with open(os.path.join(pkg, "codetable.json"), "w") as f:
        json.dump({"doc": "DSPH similarity threshold by [code length][ceil(log2(numclass))]; "
                          "numbers extracted from the reference's train/DSPH/codetable.xlsx",
                   "rows": rows}, f)
    print("wrote codetable.json", len(rows), "rows")


GENS = dict(clip_tiny=gen_clip_tiny, clip_vitb32=gen_clip_vitb32, baseclip=gen_baseclip_tiny,
            loss_dsph=gen_loss_dsph, loss_dchmt=gen_loss_dchmt, map=gen_map, codetable=gen_codetable)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    install_stubs()
    os.chdir(tempfile.mkdtemp())  # reference writes ./result/log etc.; keep it out of the repo
    torch.manual_seed(0)
    todo = [s for s in a.only.split(",") if s] or list(GENS)
    for name in todo:
        print("==", name)
        GENS[name]()
