"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of MITH's HashingModel and losses
(model/MITH.py:212-453, train/MITH/hash_train.py:80-201).  Pinned by tests/golden/mith.npz
(tests/golden/make_golden3.py runs the reference).  Reference layouts: tokens [L,N,D], concepts [K,N,D]."""
import math

import numpy as np

from .clip_oracle import F32, l2_normalize, layer_norm, linear, resblock


def gelu(x):
    erf = np.vectorize(math.erf, otypes=[np.float64])
    return (0.5 * x * (1.0 + erf(x.astype(np.float64) / math.sqrt(2.0)))).astype(F32)


def residual_mlps(x, sd, prefix, layers=2, activation="gelu"):
    """model/MITH.py:215-246."""
    for i in range(layers):
        h = layer_norm(x, sd[f"{prefix}lns.{i}.weight"], sd[f"{prefix}lns.{i}.bias"])
        u = linear(h, sd[f"{prefix}mlps.{i}.0.weight"], sd[f"{prefix}mlps.{i}.0.bias"])
        u = gelu(u) if activation == "gelu" else np.maximum(u, 0)
        x = (x + linear(u, sd[f"{prefix}mlps.{i}.3.weight"], sd[f"{prefix}mlps.{i}.3.bias"])).astype(F32)
    return x


def gcl(x, sd, prefix="gcl_i."):
    """GlobalConceptLearning (model/MITH.py:296-314): (mlp(x), tanh(x W^T))."""
    r = residual_mlps(x, sd, prefix + "mlp.")
    return r, np.tanh(r @ sd[prefix + "common_concept_embedding.weight"].T).astype(F32)


def lta(x, sim, key_padding_mask=None, top_k=8):
    """LocalizedTokenAggregation.forward (model/MITH.py:343-376).  x [L,N,D], sim [L,N,K] -> [K,N,D]."""
    sim = sim.astype(F32).copy()
    if key_padding_mask is not None:
        sim = sim + np.where(key_padding_mask, -np.inf, 0.0).T[:, :, None]
    sim = np.where(sim > 0, sim, -np.inf)
    srt = -np.sort(-sim, axis=-1)[..., :top_k]
    vmin = srt.min(-1, keepdims=True)
    sim = np.where(sim >= vmin, sim, -np.inf)
    with np.errstate(invalid="ignore"):
        m = sim.max(0, keepdims=True)
        e = np.exp(sim - m)
        w = e / e.sum(0, keepdims=True)
    w = np.where(np.isnan(w), 0.0, w).astype(F32)
    return np.einsum("lnk,lnd->knd", w, x).astype(F32)


def positional(k_bits, d):
    pe = np.zeros((k_bits, d), F32)
    pos = np.arange(k_bits, dtype=F32)[:, None]
    div = np.exp(np.arange(0, d, 2).astype(F32) * F32(-math.log(10000.0) / d))
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return (pe / F32(d ** 0.5)).astype(F32)


def lct(x, sim, sd, prefix, key_padding_mask=None, layers=2, top_k=8):
    """LocalConceptTransforming.forward (:388-396) -> (hash [N,K], transformed [K,N,D])."""
    c = lta(x, sim, key_padding_mask, top_k)                        # [K,N,D]
    K, N, D = c.shape
    c = c + positional(K, D)[:, None, :]
    y = c.transpose(1, 0, 2)                                        # batch-major for the block restatement
    for i in range(layers):
        y = resblock(y, sd, f"{prefix}transformer.resblocks.{i}.", D // 64, causal=False)
    w = np.concatenate([sd[f"{prefix}hashing.fc_list.{k}.weight"] for k in range(K)], 0)      # [K,D]
    b = np.concatenate([sd[f"{prefix}hashing.fc_list.{k}.bias"] for k in range(K)], 0)
    h = np.tanh(np.einsum("nkd,kd->nk", y, w) + b).astype(F32)
    return h, y.transpose(1, 0, 2)


def hashing_model(sd, img_tokens, txt_tokens, img_cls, txt_eos, key_padding_mask):
    """HashingModel.forward (model/MITH.py:427-453)."""
    out = {}
    ri, out["img_cls_hash"] = gcl(img_cls, sd)
    rt, out["txt_cls_hash"] = gcl(txt_eos, sd)
    out["res_img_cls"], out["res_txt_cls"] = l2_normalize(ri), l2_normalize(rt)
    hi, ti = lct(img_tokens, gcl(img_tokens, sd)[1], sd, "lct_i.")
    ht, tt = lct(txt_tokens, gcl(txt_tokens, sd)[1], sd, "lct_t.", key_padding_mask)
    out["img_tokens_hash"], out["txt_tokens_hash"] = hi, ht
    nz = lambda v: (v / np.maximum(np.sqrt((v * v).sum(-1, keepdims=True)), 1e-12)).astype(F32)
    out["trans_tokens_i"] = nz(linear(ti, sd["img_concept_proj.weight"], sd["img_concept_proj.bias"]))
    out["trans_tokens_t"] = nz(linear(tt, sd["txt_concept_proj.weight"], sd["txt_concept_proj.bias"]))
    return out


# ---------------------------------------------------------------- losses (train/MITH/hash_train.py)
def bayesian_loss(a, b, label_sim):
    s = 0.5 * np.clip(a.astype(np.float64) @ b.astype(np.float64).T, -64, 64)
    return float(-np.mean(label_sim * s - np.log(1 + np.exp(s))))


def _ce_diag(scores):
    m = scores.max(1, keepdims=True)
    lse = np.log(np.exp(scores - m).sum(1)) + m[:, 0]
    return float(np.mean(lse - np.diag(scores)))


def info_nce(o1, o2, temperature=0.07):
    s = (o1.astype(np.float64) @ o2.astype(np.float64).T) / temperature
    return 0.5 * (_ce_diag(s) + _ce_diag(s.T))


def info_nce_bmm(o1, o2, temperature=0.07):
    a, b = o1.transpose(1, 0, 2).astype(np.float64), o2.transpose(1, 0, 2).astype(np.float64)
    sim = np.einsum("nld,nmd->nlm", a, b) / temperature
    l1 = np.mean([_ce_diag(s) for s in sim])
    l2 = np.mean([_ce_diag(s.T) for s in sim])
    return 0.5 * (l1 + l2)


def compute_loss(out, label, train_labels, banks, hp, k_bits):
    """compute_loss (:149-201) + the B codes of train_epoch (:80-83).  banks: dict img_tokens,img_cls,txt_tokens,txt_cls."""
    ls = (train_labels @ label.T > 0).astype(np.float64)
    ic, tc, it, tt = out["img_cls_hash"], out["txt_cls_hash"], out["img_tokens_hash"], out["txt_tokens_hash"]
    lam = hp["hyper_lambda"]
    B = np.sign((ic * lam + it * (1 - lam)) + (tc * lam + tt * (1 - lam)))
    L = {}
    L["tokens_intra_likelihood"] = hp["hyper_tokens_intra"] * (bayesian_loss(banks["img_tokens"], it, ls) + bayesian_loss(banks["txt_tokens"], tt, ls))
    L["cls_inter_likelihood"] = hp["hyper_cls_inter"] * (bayesian_loss(banks["img_cls"], tc, ls) + bayesian_loss(banks["txt_cls"], ic, ls))
    Hi, Ht = ic * 0.5 + it * 0.5, tc * 0.5 + tt * 0.5
    q = lambda h: float(((h.astype(np.float64) - B) ** 2).sum()) / h.shape[0] / k_bits
    L["quantization"] = hp["hyper_quan"] * (q(Hi) + q(Ht))
    L["infoNCE"] = hp["hyper_info_nce"] * (info_nce(out["res_img_cls"], out["res_txt_cls"]) +
                                           hp["hyper_alpha"] * info_nce_bmm(out["trans_tokens_i"], out["trans_tokens_t"]))
    sq = lambda a, b: float(((a.astype(np.float64) - b) ** 2).sum())
    item = sq(ic, it) + sq(tc, tt)
    L["distillation"] = hp["hyper_distill"] * (item + 0.1 * item) / ic.shape[0]
    return L
