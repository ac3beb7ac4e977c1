"""ctypes wrappers of the MITH entry points of libcmh.so (signatures registered in cmh_native.SIGNATURES)."""
import ctypes as C

import torch

import cmh_native as N

EPI_BIAS, EPI_QUICKGELU, EPI_RESIDUAL, EPI_OUT_BF16, EPI_GELU, EPI_RELU = 1, 2, 4, 8, 16, 32


_bf16_weights = {}


def weight_bf16(w):
    """bf16 copy of a GEMM weight, re-cast when the parameter changes (data_ptr / _version)"""
    key = id(w)
    hit = _bf16_weights.get(key)
    if hit is None or hit[0] != (w.data_ptr(), w._version):
        hit = ((w.data_ptr(), w._version), N.cast_bf16(w.detach()))
        _bf16_weights[key] = hit
    return hit[1]


def gemm(x, w, bias=None, residual=None, act=None, dtype=N.F32, out_bf16=False):
    """out = act(x @ w.T + bias) (+ residual); act in {None,'gelu','relu','quickgelu'}.  dtype F32: exact-fp32 MFMA path on the
    f32 tensors; BF16: the operands are cast to bf16 (weights cached; an x that is bf16 already is taken as it is), accumulation /
    bias / residual stay f32.  out_bf16 (BF16 only): the result leaves the epilogue as bf16 - the next GEMM's operand - instead of f32
    followed by a cast pass: the same bits (one round-to-nearest-even of the same f32 value either way)."""
    N.require_gpu(x, w, bias, residual)
    if out_bf16 and dtype != N.BF16:
        raise N.NativeError("gemm: out_bf16 needs dtype BF16")
    if dtype == N.BF16:
        x, w = (x.contiguous() if x.dtype == torch.bfloat16 else N.cast_bf16(x)), weight_bf16(w)
    else:
        x, w = N.f32c(x), N.f32c(w)
    M, K = x.shape
    Nn = w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or (bias is not None and bias.numel() != Nn) or \
            (residual is not None and tuple(residual.shape) != (M, Nn)):
        raise N.NativeError(f"gemm: x {tuple(x.shape)}, w {tuple(w.shape)}, bias {None if bias is None else tuple(bias.shape)}, "
                            f"residual {None if residual is None else tuple(residual.shape)} do not fit together")
    out = torch.empty(M, Nn, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_RESIDUAL if residual is not None else 0) | (EPI_OUT_BF16 if out_bf16 else 0)
    epi |= {None: 0, "gelu": EPI_GELU, "relu": EPI_RELU, "quickgelu": EPI_QUICKGELU}[act]
    N.check(N.lib().cmh_linear_gemm(dtype, N.ptr(x), N.ptr(w), N.ptr(None if bias is None else N.f32c(bias)),
                                    N.ptr(None if residual is None else N.f32c(residual)), N.ptr(out), M, Nn, K, epi,
                                    N.stream_ptr(x.device)), "cmh_linear_gemm")
    return out


def vit_encode_tokens(clip, image):
    image = N.f32c(image)
    N.require_gpu(image)
    s = clip._vit_struct()
    B = image.shape[0]
    T = (s.resolution // s.patch) ** 2 + 1
    out = torch.empty(B * T, s.embed_dim, dtype=torch.float32, device=image.device)
    ws = N.workspace(N.lib().cmh_vit_workspace_bytes(C.byref(s), B), image.device, f"vit@{N.stream_ptr(image.device)}")      # one scratch per stream: the eval loops run batches on alternating streams
    N.check(N.lib().cmh_vit_encode_tokens(C.byref(s), N.ptr(image), B, N.ptr(out), N.ptr(ws), ws.numel(),
                                          N.stream_ptr(image.device)), "cmh_vit_encode_tokens")
    return out.view(B, T, s.embed_dim)


def text_encode_tokens(clip, text, key_padding_mask, padded_unused=False):
    """padded_unused: the caller reads no padded position of the result (MITH: HashingModel masks them) - they are then not computed
    and come back as zeros (cmh_text_encode_tokens_packed)"""
    N.require_gpu(text)
    text = text.to(torch.int64).contiguous()
    s = clip._text_struct()
    B, L = text.shape
    out = torch.empty(B * L, s.embed_dim, dtype=torch.float32, device=text.device)
    rows = torch.empty(B, dtype=torch.int32, device=text.device)
    kpm = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
    ws = N.workspace(N.lib().cmh_text_workspace_bytes(C.byref(s), B, L), text.device, f"text@{N.stream_ptr(text.device)}")
    fn = N.lib().cmh_text_encode_tokens_packed if padded_unused and kpm is not None else N.lib().cmh_text_encode_tokens
    N.check(fn(C.byref(s), N.ptr(text), B, L, N.ptr(kpm), N.ptr(out), N.ptr(rows), N.ptr(ws), ws.numel(), N.stream_ptr(text.device)),
            "cmh_text_encode_tokens")
    return out.view(B, L, s.embed_dim), rows


def transformer_blocks(block_array, layers, x, B, T, dtype=N.F32):
    """x f32 [B*T, d] -> same shape, after `layers` ResidualAttentionBlocks (no mask); block_array's GEMM weights in `dtype`."""
    x = N.f32c(x).clone()
    d = x.shape[1]
    ws = N.workspace(N.lib().cmh_blocks_workspace_bytes(dtype, B, T, d), x.device, f"blocks@{N.stream_ptr(x.device)}")
    N.check(N.lib().cmh_transformer_blocks(C.cast(block_array, C.POINTER(N.BlockWeights)), layers, dtype, N.ptr(x), B, T,
                                           d, 0, None, N.ptr(ws), ws.numel(), N.stream_ptr(x.device)),
            "cmh_transformer_blocks")
    return x


def lta(tokens, sim, key_padding_mask, l0, L, top_k):
    """tokens f32 [B, Ltot, D], sim f32 [B, Ltot, K] -> merged concepts [B, K, D]."""
    tokens, sim = N.f32c(tokens), N.f32c(sim)
    B, Ltot, D = tokens.shape
    K = sim.shape[2]
    N.fit("lta", (sim, (B, Ltot, K)), (key_padding_mask, (B, L)))
    out = torch.empty(B, K, D, dtype=torch.float32, device=tokens.device)
    kpm = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
    N.check(N.lib().cmh_mith_lta(N.ptr(tokens), N.ptr(sim), N.ptr(kpm), N.ptr(out), B, Ltot, l0, L, K, D, top_k,
                                 N.stream_ptr(tokens.device)), "cmh_mith_lta")
    return out


def add_positional(x, pe):
    x = N.f32c(x).clone()
    B, T, D = x.shape
    N.check(N.lib().cmh_add_positional(N.ptr(x), N.ptr(N.f32c(pe)), B, T, D, N.stream_ptr(x.device)), "cmh_add_positional")
    return x


def bitwise_hash(x, w, bias):
    x = N.f32c(x)
    B, K, D = x.shape
    N.fit("bitwise_hash", (w, (K, D)), (bias, (K,)))
    out = torch.empty(B, K, dtype=torch.float32, device=x.device)
    N.check(N.lib().cmh_bitwise_hash(N.ptr(x), N.ptr(N.f32c(w)), N.ptr(N.f32c(bias)), N.ptr(out), B, K, D,
                                     N.stream_ptr(x.device)), "cmh_bitwise_hash")
    return out


def l2_normalize_rows(x):
    x = N.f32c(x)
    y = torch.empty_like(x)
    R = x.numel() // x.shape[-1]
    N.check(N.lib().cmh_l2_normalize_rows(N.ptr(x), N.ptr(y), R, x.shape[-1], N.stream_ptr(x.device)), "cmh_l2_normalize_rows")
    return y


def mith_mix(ic, it, tc, tt, lam):
    ic, it, tc, tt = (N.f32c(t) for t in (ic, it, tc, tt))
    N.fit("mith_mix", (it, ic.shape), (tc, ic.shape), (tt, ic.shape))
    Bc, Hi, Ht = torch.empty_like(ic), torch.empty_like(ic), torch.empty_like(ic)
    N.check(N.lib().cmh_mith_mix(N.ptr(ic), N.ptr(it), N.ptr(tc), N.ptr(tt), float(lam), N.ptr(Bc), N.ptr(Hi), N.ptr(Ht),
                                 ic.numel(), N.stream_ptr(ic.device)), "cmh_mith_mix")
    return Bc, Hi, Ht


def _scalar(dev):
    return torch.empty(1, dtype=torch.float32, device=dev)


def sq_diff_sum(a, b):
    a, b = N.f32c(a), N.f32c(b)
    N.fit("sq_diff_sum", (b, a.shape))
    out, ws = _scalar(a.device), N.workspace(256, a.device, f"loss@{N.stream_ptr(a.device)}")
    N.check(N.lib().cmh_sq_diff_sum(N.ptr(a), N.ptr(b), a.numel(), N.ptr(out), N.ptr(ws), ws.numel(), N.stream_ptr(a.device)),
            "cmh_sq_diff_sum")
    return out[0]


def bayesian_loss(bank, batch, bank_label, label):
    bank, batch, bank_label, label = (N.f32c(t) for t in (bank, batch, bank_label, label))
    N.fit("bayesian_loss", (batch, (batch.shape[0], bank.shape[1])), (bank_label, (bank.shape[0], label.shape[1])),
          (label, (batch.shape[0], label.shape[1])))
    out, ws = _scalar(bank.device), N.workspace(256, bank.device, f"loss@{N.stream_ptr(bank.device)}")
    N.check(N.lib().cmh_mith_bayesian_loss(N.ptr(bank), N.ptr(batch), N.ptr(bank_label), N.ptr(label), bank.shape[0],
                                           batch.shape[0], bank.shape[1], label.shape[1], N.ptr(out), N.ptr(ws), ws.numel(),
                                           N.stream_ptr(bank.device)), "cmh_mith_bayesian_loss")
    return out[0]


def info_nce(a, b, group=None, temperature=0.07):
    a, b = N.f32c(a), N.f32c(b)
    N.fit("info_nce", (b, a.shape))
    D = a.shape[-1]
    R = a.numel() // D
    G = R if group is None else group
    out, ws = _scalar(a.device), N.workspace(N.lib().cmh_info_nce_workspace_bytes(R, G), a.device, f"nce@{N.stream_ptr(a.device)}")
    N.check(N.lib().cmh_info_nce(N.ptr(a), N.ptr(b), R, G, D, float(temperature), N.ptr(out), N.ptr(ws), ws.numel(),
                                 N.stream_ptr(a.device)), "cmh_info_nce")
    return out[0]
