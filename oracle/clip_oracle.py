"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

numpy fp32 restatement of the reference's encode -> hash -> loss path.  Every function
cites the reference lines it follows (paths relative to /root/reference).  Pinned against
outputs of the reference itself: tests/golden/make_golden.py -> tests/golden/*.npz
(tests/test_oracle_*.py).

Layout: activations are [B, T, d] (batch-major); the reference permutes to [T, B, d] (LND)
around its transformers (model/base/model.py:241-246, :363-365) which changes no value.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- primitives
def layer_norm(x, w, b, eps=1e-5):
    """model/base/model.py:153-159 — nn.LayerNorm computed in fp32, biased variance."""
    x = x.astype(F32)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * w + b).astype(F32)


def quick_gelu(x):
    """model/base/model.py:162-164 — x * sigmoid(1.702 x)."""
    return (x / (F32(1) + np.exp(-F32(1.702) * x))).astype(F32)


def linear(x, w, b=None):
    y = x @ w.T
    return y + b if b is not None else y


def softmax(x, axis=-1):
    m = x.max(axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis, keepdims=True)


def mha(x, in_w, in_b, out_w, out_b, heads, causal, key_padding_mask=None):
    """nn.MultiheadAttention(d, heads)(x,x,x, need_weights=False, attn_mask) as used at
    model/base/model.py:171,184-189: packed in_proj [3d,d], head dim d/heads, q scaled by
    1/sqrt(hd), additive -inf causal mask (:340-346) for the text tower."""
    B, T, d = x.shape
    hd = d // heads
    qkv = linear(x, in_w, in_b)                                   # [B,T,3d]
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, T, heads, hd).transpose(0, 2, 1, 3)
               for i in range(3))                                 # [B,h,T,hd]
    s = (q * F32(hd ** -0.5)) @ k.transpose(0, 1, 3, 2)           # [B,h,T,T]
    if causal:
        s = s + np.triu(np.full((T, T), -np.inf, F32), 1)
    if key_padding_mask is not None:                               # bool [B,T], True = ignore
        s = np.where(key_padding_mask[:, None, None, :], -np.inf, s)
    p = softmax(s.astype(F32), -1).astype(F32)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
    return linear(o, out_w, out_b).astype(F32)


def resblock(x, sd, prefix, heads, causal, key_padding_mask=None):
    """model/base/model.py:167-196 ResidualAttentionBlock.forward (:191-196)."""
    g = lambda k: sd[prefix + k]
    h = layer_norm(x, g("ln_1.weight"), g("ln_1.bias"))
    x = x + mha(h, g("attn.in_proj_weight"), g("attn.in_proj_bias"), g("attn.out_proj.weight"),
                g("attn.out_proj.bias"), heads, causal, key_padding_mask)
    h = layer_norm(x, g("ln_2.weight"), g("ln_2.bias"))
    h = quick_gelu(linear(h, g("mlp.c_fc.weight"), g("mlp.c_fc.bias")))
    return (x + linear(h, g("mlp.c_proj.weight"), g("mlp.c_proj.bias"))).astype(F32)


def _layers(sd, prefix):
    n = 0
    while f"{prefix}.resblocks.{n}.ln_1.weight" in sd:
        n += 1
    return n


# ----------------------------------------------------------------------------- encoders
def patchify(image, p):
    """conv1 with kernel=stride=p, no bias (model/base/model.py:215,231-235) is a GEMM over
    non-overlapping patches; rows ordered (gy, gx), columns ordered (c, py, px)."""
    B, C, H, W = image.shape
    g = H // p
    x = image.reshape(B, C, g, p, g, p).transpose(0, 2, 4, 1, 3, 5)
    return np.ascontiguousarray(x.reshape(B, g * g, C * p * p))


def encode_image_tokens(sd, image, taps=None):
    """VisionTransformer.forward model/base/model.py:228-245 up to (and incl.) the blocks."""
    w = sd["visual.conv1.weight"]
    vw, p = w.shape[0], w.shape[-1]
    x = patchify(image.astype(F32), p) @ w.reshape(vw, -1).T       # [B,g*g,vw]
    cls = np.broadcast_to(sd["visual.class_embedding"], (x.shape[0], 1, vw))
    x = np.concatenate([cls, x], 1) + sd["visual.positional_embedding"]
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    if taps is not None:
        taps["v_ln_pre"] = x
    heads = vw // 64                                               # :284
    for i in range(_layers(sd, "visual.transformer")):
        x = resblock(x, sd, f"visual.transformer.resblocks.{i}.", heads, causal=False)
        if taps is not None:
            taps[f"v_block{i}"] = x
    return x


def encode_image(sd, image, taps=None):
    """model/base/model.py:247-252 — ln_post on the class token, then @ proj."""
    x = encode_image_tokens(sd, image, taps)
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return (x @ sd["visual.proj"]).astype(F32)


def encode_text(sd, text, taps=None):
    """CLIP.encode_text model/base/model.py:359-372: token_embedding gather, + positional
    [:L], causal blocks, ln_final, row at argmax(token id) (EOT is the largest id), @ text_projection."""
    B, L = text.shape
    tw = sd["ln_final.weight"].shape[0]
    x = (sd["token_embedding.weight"][text] + sd["positional_embedding"][:L]).astype(F32)
    heads = tw // 64                                               # :437
    for i in range(_layers(sd, "transformer")):
        x = resblock(x, sd, f"transformer.resblocks.{i}.", heads, causal=True)
        if taps is not None:
            taps[f"t_block{i}"] = x
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    eot = text.argmax(-1)
    return (x[np.arange(B), eot] @ sd["text_projection"]).astype(F32)


# ----------------------------------------------------------------------------- heads
def linear_hash(feat, w, b, drop_mask=None, p=0.2):
    """model/modelbase.py:25-35 LinearHash: tanh(dropout_{0.2}(fc(x))).  eval -> no mask;
    training parity needs the mask injected (inverted-dropout scaling 1/(1-p))."""
    y = linear(feat, w, b)
    if drop_mask is not None:
        y = y * drop_mask / F32(1 - p)
    return np.tanh(y).astype(F32)


def dchmt_hash_layer(feat, fc_w, fc_b, bits_w, bits_b):
    """model/DCHMT.py:8-26 HashLayer: relu(fc 512->128) then K x softmax(Linear(128->2)).
    bits_w [2K,128] holds the K two-row Linears stacked in order. Returns [B,K,2]."""
    e = np.maximum(linear(feat, fc_w, fc_b), 0)
    z = linear(e, bits_w, bits_b).reshape(feat.shape[0], -1, 2)
    return softmax(z.astype(F32), -1).astype(F32)


def sign_codes(h):
    """train/base.py:141-143 — torch.sign (0 stays 0)."""
    return np.sign(h).astype(F32)


def dchmt_codes(pairs):
    """train/base.py:150-158 make_hash_code_DCHMT: argmax over the pair; index 0 -> -1, 1 -> +1
    (argmax returns the first maximum, so a tie gives -1)."""
    return np.where(pairs.argmax(-1) == 0, -1.0, 1.0).astype(F32)


def pre_layer(feat, w, b):
    """model/DNPH_TOMM.py:7-14 Pre_Layer."""
    return linear(feat, w, b).astype(F32)


# ----------------------------------------------------------------------------- losses
def l2_normalize(x, eps=1e-12):
    """F.normalize(x, p=2, dim=1)."""
    n = np.sqrt((x * x).sum(1, keepdims=True))
    return (x / np.maximum(n, eps)).astype(F32)


def calc_neighbor(a, b):
    """utils/calc_utils.py:42-45 / utils/utils.py:26-28."""
    return ((a @ b.T) > 0).astype(F32)


def dsph_hyp_loss(x, y, label, proxies, threshold, alpha):
    """train/DSPH/loss.py:22-72 HyP.forward."""
    x, y, label, proxies = (np.asarray(a, F32) for a in (x, y, label, proxies))
    pn = l2_normalize(proxies)
    cos, cos_t = l2_normalize(x) @ pn.T, l2_normalize(y) @ pn.T
    thr = F32(threshold)
    P, N = label != 0, label == 0
    p_num, n_num = int(P.sum()), int(N.sum())
    is_one = label == 1
    tot = np.float64(0)
    for c in (cos, cos_t):
        tot += np.where(is_one, 1 - c, 0).sum(dtype=np.float64) / p_num
        tot += np.where(N, np.maximum(c - thr, 0), 0).sum(dtype=np.float64) / n_num
    if alpha > 0:
        idx = label.sum(1) > 1
        lab_, x_, t_ = label[idx], l2_normalize(x[idx]), l2_normalize(y[idx])
        zero = (lab_ @ lab_.T) == 0
        nz = int(zero.sum())
        if nz > 0:
            for s in (x_ @ x_.T, t_ @ t_.T, x_ @ t_.T):
                tot += (F32(alpha) * np.where(zero, np.maximum(s - thr, 0), 0)).sum(dtype=np.float64) / nz
    return F32(tot)


def cdist(a, b):
    """torch.cdist(a, b, p=2)."""
    d = a[:, None, :].astype(np.float64) - b[None, :, :].astype(np.float64)
    return np.sqrt((d * d).sum(-1)).astype(F32)


def cosine_similarity(a, b):
    """utils/utils.py:58-69 (row-normalise unless the whole matrix is zero)."""
    if np.any(a != 0):
        a = a / np.sqrt((a * a).sum(-1, keepdims=True))
    if np.any(b != 0):
        b = b / np.sqrt((b * b).sum(-1, keepdims=True))
    return (a @ b.T).astype(F32)


def dchmt_similarity_loss(a, b, label_sim, K, fn="euclidean", loss_type="l2", vartheta=0.5,
                          sim_threshold=0.1):
    """train/DCHMT/hash_train.py:82-114."""
    threshold = sim_threshold if sim_threshold != 0 else 0.05
    sim = (1 - cosine_similarity(a, b)) if fn == "cosine" else cdist(a, b)
    pos = sim * label_sim
    neg = sim * (1 - label_sim)
    if fn == "cosine":
        pos = np.maximum(pos, F32(threshold)) - F32(threshold)
        neg = np.minimum(neg, F32(1.0))
        neg = (1 - label_sim) - neg
    else:
        mx = F32(float(K * 2 * vartheta) ** 0.5)
        neg = np.minimum(neg, mx)
        neg = mx * (1 - label_sim) - neg
    if loss_type == "l1":
        return pos.mean(dtype=np.float64), neg.mean(dtype=np.float64)
    return (pos.astype(np.float64) ** 2).mean(), (neg.astype(np.float64) ** 2).mean()


def dchmt_our_loss(img, txt, label, K, fn="euclidean", loss_type="l2", vartheta=0.5,
                   sim_threshold=0.1):
    """train/DCHMT/hash_train.py:116-150 with hash_layer == 'select' (the default)."""
    ls = calc_neighbor(label, label)
    tot = 0.0
    for a, b in ((img, txt), (img, img), (txt, txt)):
        p, n = dchmt_similarity_loss(a, b, ls, K, fn, loss_type, vartheta, sim_threshold)
        tot += p + n
    return F32(tot)
