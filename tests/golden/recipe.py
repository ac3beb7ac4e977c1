"""Seeded input/weight recipes shared by the golden generator and the tests.

Own code (no reference imports).  Everything is drawn from numpy PCG64 streams so that
the generating script (which pushes the arrays into the *reference* modules with
load_state_dict) and the tests (which feed the oracle and the HIP path) see bit-identical
weights and inputs without committing megabytes of weights.  Fixtures therefore hold only
expected OUTPUTS (plus small crafted inputs).

state_dict key names/shapes follow the reference's CLIP (model/base/model.py:255-309,
SURVEY.md §8b "state_dict contract").
"""
from __future__ import annotations

import zlib

import numpy as np

SOT, EOT = 49406, 49407

CLIP_TINY = dict(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128,
                 vision_patch_size=32, context_length=16, vocab_size=512,
                 transformer_width=128, transformer_heads=2, transformer_layers=2)
CLIP_VITB32 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                   vision_patch_size=32, context_length=77, vocab_size=49408,
                   transformer_width=512, transformer_heads=8, transformer_layers=12)


def _rng(seed: int, tag: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(tag.encode())]))


def clip_state_shapes(cfg: dict) -> dict:
    """Ordered {key: shape} for the reference CLIP state_dict of a ViT config."""
    d, vw, tw = cfg["embed_dim"], cfg["vision_width"], cfg["transformer_width"]
    p = cfg["vision_patch_size"]
    grid = cfg["image_resolution"] // p
    shapes = {
        "positional_embedding": (cfg["context_length"], tw),
        "text_projection": (tw, d),
        "logit_scale": (),
        "visual.class_embedding": (vw,),
        "visual.positional_embedding": (grid * grid + 1, vw),
        "visual.proj": (vw, d),
        "visual.conv1.weight": (vw, 3, p, p),
        "visual.ln_pre.weight": (vw,), "visual.ln_pre.bias": (vw,),
        "visual.ln_post.weight": (vw,), "visual.ln_post.bias": (vw,),
        "token_embedding.weight": (cfg["vocab_size"], tw),
        "ln_final.weight": (tw,), "ln_final.bias": (tw,),
    }
    for prefix, w, layers in (("visual.transformer", vw, cfg["vision_layers"]),
                              ("transformer", tw, cfg["transformer_layers"])):
        for i in range(layers):
            b = f"{prefix}.resblocks.{i}."
            shapes[b + "attn.in_proj_weight"] = (3 * w, w)
            shapes[b + "attn.in_proj_bias"] = (3 * w,)
            shapes[b + "attn.out_proj.weight"] = (w, w)
            shapes[b + "attn.out_proj.bias"] = (w,)
            shapes[b + "ln_1.weight"] = (w,)
            shapes[b + "ln_1.bias"] = (w,)
            shapes[b + "mlp.c_fc.weight"] = (4 * w, w)
            shapes[b + "mlp.c_fc.bias"] = (4 * w,)
            shapes[b + "mlp.c_proj.weight"] = (w, 4 * w)
            shapes[b + "mlp.c_proj.bias"] = (w,)
            shapes[b + "ln_2.weight"] = (w,)
            shapes[b + "ln_2.bias"] = (w,)
    return shapes


def _scale_for(key: str, shape: tuple, cfg: dict) -> tuple[float, float]:
    """(offset, std) per tensor, roughly the reference's init distributions
    (model/base/model.py:219-226,311-338) but with non-zero biases / LN noise so every
    term of the path is exercised."""
    width = cfg["vision_width"] if key.startswith("visual.") else cfg["transformer_width"]
    layers = cfg["vision_layers"] if key.startswith("visual.") else cfg["transformer_layers"]
    if key == "logit_scale":
        return float(np.log(1 / 0.07)), 0.0
    if key.endswith("ln_1.weight") or key.endswith("ln_2.weight") or key in (
            "visual.ln_pre.weight", "visual.ln_post.weight", "ln_final.weight"):
        return 1.0, 0.1
    if ".ln_" in key or key.startswith("ln_final") or "ln_pre" in key or "ln_post" in key:
        return 0.0, 0.05
    if key.endswith("bias"):
        return 0.0, 0.02
    if key == "token_embedding.weight":
        return 0.0, 0.02
    if key == "positional_embedding":
        return 0.0, 0.01
    if key in ("visual.class_embedding", "visual.positional_embedding", "visual.proj",
               "text_projection"):
        return 0.0, width ** -0.5
    if key == "visual.conv1.weight":
        return 0.0, float(np.prod(shape[1:])) ** -0.5
    if key.endswith("attn.in_proj_weight"):
        return 0.0, width ** -0.5
    if key.endswith("attn.out_proj.weight") or key.endswith("mlp.c_proj.weight"):
        return 0.0, (width ** -0.5) * ((2 * layers) ** -0.5)
    if key.endswith("mlp.c_fc.weight"):
        return 0.0, (2 * width) ** -0.5
    raise KeyError(key)


def fp16_round(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).astype(np.float32)


def is_fp16_converted(key: str) -> bool:
    """Tensors the reference rounds through fp16 when a checkpoint goes through
    build_model (model/base/model.py:391-412 convert_weights + :453 load_state_dict):
    Conv/Linear weight+bias, MHA in_proj/bias, visual.proj, text_projection."""
    if key in ("visual.proj", "text_projection", "visual.conv1.weight"):
        return True
    if ".ln_" in key or key.startswith("ln_final") or "ln_pre" in key or "ln_post" in key:
        return False
    return ".attn." in key or ".mlp." in key


def clip_state_dict(cfg: dict, seed: int, fp16_roundtrip: bool = False) -> dict:
    """{key: float32 ndarray} in sorted-key draw order (one PCG64 stream per tensor)."""
    out = {}
    for key, shape in sorted(clip_state_shapes(cfg).items()):
        off, std = _scale_for(key, shape, cfg)
        a = np.asarray(off + std * _rng(seed, key).standard_normal(shape), dtype=np.float32)
        if fp16_roundtrip and is_fp16_converted(key):
            a = fp16_round(a)
        out[key] = a
    return out


def images(batch: int, resolution: int, seed: int) -> np.ndarray:
    """CLIP-normalised images are ~N(0,1)."""
    return _rng(seed, "image").standard_normal((batch, 3, resolution, resolution)).astype(np.float32)


def captions(batch: int, length: int, vocab: int, seed: int) -> np.ndarray:
    """[B, L] int64: SOT, body, EOT(=vocab-1, the max id), 0-padding (dataset/base.py:64-81).
    For the real vocab EOT is 49407; for tiny vocabularies the top id plays that role, the
    reference only relies on argmax (model/base/model.py:370)."""
    rng = _rng(seed, "caption")
    eot, sot = vocab - 1, vocab - 2
    t = np.zeros((batch, length), np.int64)
    for i in range(batch):
        n = int(rng.integers(2, length))  # position of EOT, 2..L-1
        if i == 0:
            n = length - 1  # one full-length caption
        t[i, 0] = sot
        t[i, 1:n] = rng.integers(1, sot, size=n - 1)
        t[i, n] = eot
    return t


def labels(n: int, classes: int, seed: int, p: float = 0.15, tag: str = "label") -> np.ndarray:
    """Bernoulli(p) multi-hot labels, all-zero rows allowed (exercise the skip path)."""
    return (_rng(seed, tag).random((n, classes)) < p).astype(np.float32)


def sign_codes(n: int, bits: int, seed: int, tag: str, zeros: int = 0) -> np.ndarray:
    """+-1 codes; `zeros` entries forced to exact 0 (sign(0)=0 can occur)."""
    rng = _rng(seed, tag)
    c = np.where(rng.random((n, bits)) < 0.5, -1.0, 1.0).astype(np.float32)
    if zeros:
        r = rng.integers(0, n, size=zeros)
        b = rng.integers(0, bits, size=zeros)
        c[r, b] = 0.0
    return c


def correlated_codes(lab: np.ndarray, bits: int, seed: int, tag: str, noise: float = 0.5) -> np.ndarray:
    """sign(labels.W + noise*randn): codes correlated with labels (mAP well above chance)."""
    rng = _rng(seed, tag)
    w = _rng(seed, "codeW%d" % bits).standard_normal((lab.shape[1], bits))
    x = lab @ w + noise * rng.standard_normal((lab.shape[0], bits))
    return np.where(x > 0, 1.0, -1.0).astype(np.float32)


def head_linear(in_dim: int, out_dim: int, seed: int, tag: str) -> tuple[np.ndarray, np.ndarray]:
    """weight [out,in] ~ kaiming-uniform(fan_out)-like, bias small non-zero."""
    rng = _rng(seed, tag)
    bound = (6.0 / out_dim) ** 0.5
    w = rng.uniform(-bound, bound, size=(out_dim, in_dim)).astype(np.float32)
    b = (0.05 * rng.standard_normal(out_dim)).astype(np.float32)
    return w, b


def features(n: int, dim: int, seed: int, tag: str) -> np.ndarray:
    return (0.5 * _rng(seed, tag).standard_normal((n, dim))).astype(np.float32)
