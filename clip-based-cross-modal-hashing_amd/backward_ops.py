"""ctypes wrappers of the backward building blocks (include/cmh.h, csrc/backward.hip, csrc/attention_bwd.hip).  Same rules as
cmh_native: GPU tensors only, no fallback."""
import torch

import cmh_native as N

KIND = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def transpose(src, dst_dtype=None):
    """[R, C] -> [C, R], optionally cast (f32 -> bf16 for GEMM operands)."""
    N.require_gpu(src)
    src = src.contiguous()
    R, Cc = src.shape
    dst = torch.empty(Cc, R, dtype=dst_dtype or src.dtype, device=src.device)
    N.check(N.lib().cmh_transpose(N.ptr(src), KIND[src.dtype], N.ptr(dst), KIND[dst.dtype], R, Cc, N.stream_ptr(src.device)),
            "cmh_transpose")
    return dst


def colsum(x):
    N.require_gpu(x)
    x = x.contiguous()
    R, Cc = x.shape
    out = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = N.workspace(N.lib().cmh_colsum_workspace_bytes(R, Cc), x.device, "bwd")
    N.check(N.lib().cmh_colsum(N.ptr(x), KIND[x.dtype], R, Cc, N.ptr(out), N.ptr(ws), ws.numel(), N.stream_ptr(x.device)),
            "cmh_colsum")
    return out


def layernorm_backward(x, dy, gamma, dx=None):
    """-> (dx f32 [M,d] (accumulated into `dx` when given), dgamma, dbeta)"""
    N.require_gpu(x, dy, gamma, dx)
    x, dy, gamma = x.contiguous(), dy.contiguous(), N.f32c(gamma)
    M, d = x.shape
    acc = dx is not None
    if dx is None:
        dx = torch.empty(M, d, dtype=torch.float32, device=x.device)
    dg = torch.empty(d, dtype=torch.float32, device=x.device)
    db = torch.empty(d, dtype=torch.float32, device=x.device)
    ws = N.workspace(N.lib().cmh_layernorm_backward_workspace_bytes(M, d), x.device, "bwd")
    N.check(N.lib().cmh_layernorm_backward(N.ptr(x), KIND[x.dtype], N.ptr(dy), KIND[dy.dtype], N.ptr(gamma), M, d, N.ptr(dx),
                                           int(acc), N.ptr(dg), N.ptr(db), N.ptr(ws), ws.numel(), N.stream_ptr(x.device)),
            "cmh_layernorm_backward")
    return dx, dg, db


def quick_gelu(pre):
    N.require_gpu(pre)
    pre = pre.contiguous()
    out = torch.empty_like(pre)
    N.check(N.lib().cmh_quick_gelu(N.ptr(pre), N.ptr(out), pre.numel(), KIND[pre.dtype], N.stream_ptr(pre.device)),
            "cmh_quick_gelu")
    return out


def attention_backward(qkv, o, dout, B, T, causal, key_padding_mask=None):
    """-> dqkv like qkv ([B*T, 3d]); o is the forward output of cmh_attention, dout its gradient."""
    N.require_gpu(qkv, o, dout, key_padding_mask)
    qkv, o, dout = qkv.contiguous(), o.contiguous(), dout.contiguous()
    dt = N.BF16 if qkv.dtype == torch.bfloat16 else N.F32
    d = qkv.shape[1] // 3
    dqkv = torch.empty_like(qkv)
    kpm = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
    N.check(N.lib().cmh_attention_backward(dt, N.ptr(qkv), N.ptr(o), N.ptr(dout), N.ptr(dqkv), B, T, d, int(causal), N.ptr(kpm),
                                           N.stream_ptr(qkv.device)), "cmh_attention_backward")
    return dqkv


# ------------------------------------------------------------------------------------------ heads / loss (autograd bridges)
class LinearAct(torch.autograd.Function):
    """cmh_linear_act with its backward (LinearHash: fc -> dropout -> tanh, model/modelbase.py:25-35)."""

    @staticmethod
    def forward(ctx, x, w, b, act, drop_mask, p):
        y = N.linear_act(x, w, b, act, drop_mask, p)
        ctx.save_for_backward(N.f32c(x), N.f32c(w), y, None if drop_mask is None else N.f32c(drop_mask))
        ctx.act, ctx.keep = act, 1.0 / (1.0 - p)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, mask = ctx.saved_tensors
        dy = N.f32c(dy)
        M, K = x.shape
        Nn = w.shape[0]
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty(Nn, dtype=torch.float32, device=x.device)
        ws = N.workspace(M * Nn * 4 + 256, x.device, "bwd")
        N.check(N.lib().cmh_linear_act_backward(N.ptr(x), N.ptr(w), N.ptr(y), N.ptr(dy), N.ptr(mask), ctx.keep, ctx.act, N.ptr(dx),
                                                N.ptr(dw), N.ptr(db), M, Nn, K, N.ptr(ws), ws.numel(), N.stream_ptr(x.device)),
                "cmh_linear_act_backward")
        return dx, dw, db, None, None, None


class HypLoss(torch.autograd.Function):
    """cmh_dsph_hyp_loss with its backward (train/DSPH/loss.py:22-72)."""

    @staticmethod
    def forward(ctx, x, y, label, proxies, threshold, alpha):
        x, y, label, proxies = N.f32c(x), N.f32c(y), N.f32c(label), N.f32c(proxies)
        ctx.save_for_backward(x, y, label, proxies)
        ctx.threshold, ctx.alpha = float(threshold), float(alpha)
        return N.dsph_hyp_loss(x, y, label, proxies, threshold, alpha)

    @staticmethod
    def backward(ctx, dloss):
        x, y, label, proxies = ctx.saved_tensors
        B, K = x.shape
        Cn = label.shape[1]
        dx, dy, dp = torch.empty_like(x), torch.empty_like(y), torch.empty_like(proxies)
        dl = N.f32c(dloss).reshape(1)
        ws = N.workspace(N.lib().cmh_head_backward_workspace_bytes(B, K, Cn), x.device, "bwd")
        N.check(N.lib().cmh_dsph_hyp_loss_backward(N.ptr(x), N.ptr(y), N.ptr(label), N.ptr(proxies), B, K, Cn, ctx.threshold, ctx.alpha,
                                                   N.ptr(dl), N.ptr(dx), N.ptr(dy), N.ptr(dp), N.ptr(ws), ws.numel(),
                                                   N.stream_ptr(x.device)), "cmh_dsph_hyp_loss_backward")
        return dx, dy, None, dp, None, None


class LayerNormFn(torch.autograd.Function):
    """cmh_layernorm (f32 rows) with cmh_layernorm_backward: the TwDH text head's `norm` (model/TwDH.py:61,78)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = N.f32c(x)
        ctx.save_for_backward(x, N.f32c(w))
        return N.layernorm(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dg, db = layernorm_backward(x, N.f32c(dy), w)
        return dx, dg, db


class BatchNorm1dTrain(torch.autograd.Function):
    """cmh_batchnorm1d_train (batch statistics) with its backward: the TwDH image head's `norm`."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x, w = N.f32c(x), N.f32c(w)
        ctx.save_for_backward(x, w)
        ctx.eps = float(eps)
        return N.batchnorm1d_train(x, w, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = N.f32c(dy)
        B, d = x.shape
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(w)
        N.check(N.lib().cmh_batchnorm1d_backward(N.ptr(x), N.ptr(w), ctx.eps, N.ptr(dy), N.ptr(dx), N.ptr(dw), N.ptr(db), B, d,
                                                 N.stream_ptr(x.device)), "cmh_batchnorm1d_backward")
        return dx, dw, db, None


class TwdhLoss(torch.autograd.Function):
    """cmh_twdh_loss -> (nce, quan) for one code length with its backward (train/TwDH/hash_train.py:117-139)."""

    @staticmethod
    def forward(ctx, p_img, p_txt, target):
        p_img, p_txt, target = N.f32c(p_img), N.f32c(p_txt), N.f32c(target)
        ctx.save_for_backward(p_img, p_txt, target)
        nce, quan = N.twdh_loss(p_img, p_txt, target)
        return nce.clone(), quan.clone()

    @staticmethod
    def backward(ctx, d_nce, d_quan):
        p_img, p_txt, target = ctx.saved_tensors
        B, K = target.shape
        di, dt = torch.empty_like(p_img), torch.empty_like(p_txt)
        gn = None if d_nce is None else N.f32c(d_nce).reshape(1)
        gq = None if d_quan is None else N.f32c(d_quan).reshape(1)
        N.check(N.lib().cmh_twdh_loss_backward(N.ptr(p_img), N.ptr(p_txt), N.ptr(target), B, K, N.ptr(gn), N.ptr(gq), N.ptr(di), N.ptr(dt),
                                               N.stream_ptr(p_img.device)), "cmh_twdh_loss_backward")
        return di, dt, None


class DnphLoss(torch.autograd.Function):
    """cmh_dnph_loss with its backward (DNPH_out, train/DNPH_TOMM/loss.py:14-32, plus the `- 0.1 * noise_loss` term of
    train/DNPH_TOMM/hash_train.py:65-81 when the noise rows are given).  Returns loss1 without noise rows, else the step loss."""

    @staticmethod
    def forward(ctx, f1, f2, p1, p2, label, proxies, noise_1, noise_2, mrg, noise_weight):
        ts = [N.f32c(t) for t in (f1, f2, p1, p2, label, proxies)]
        n1 = None if noise_1 is None else N.f32c(noise_1)
        n2 = None if noise_2 is None else N.f32c(noise_2)
        ctx.save_for_backward(*ts, n1, n2)
        ctx.mrg, ctx.nw = float(mrg), float(noise_weight)
        total, loss1, _ = N.dnph_loss(*ts, n1, n2, mrg, noise_weight)
        return (loss1 if n1 is None else total).clone()

    @staticmethod
    def backward(ctx, dloss):
        f1, f2, p1, p2, label, proxies, n1, n2 = ctx.saved_tensors
        B, K = f1.shape
        Cn = label.shape[1]
        d1, d2, dp1, dp2, dpx = (torch.empty_like(t) for t in (f1, f2, p1, p2, proxies))
        dl = N.f32c(dloss).reshape(1)
        ws = N.workspace(N.lib().cmh_dnph_backward_workspace_bytes(B, K, Cn), f1.device, "bwd")
        N.check(N.lib().cmh_dnph_loss_backward(N.ptr(f1), N.ptr(f2), N.ptr(p1), N.ptr(p2), N.ptr(label), N.ptr(proxies), N.ptr(n1),
                                               N.ptr(n2), B, K, Cn, ctx.mrg, ctx.nw, N.ptr(dl), N.ptr(d1), N.ptr(d2), N.ptr(dp1),
                                               N.ptr(dp2), N.ptr(dpx), N.ptr(ws), ws.numel(), N.stream_ptr(f1.device)),
                "cmh_dnph_loss_backward")
        return d1, d2, dp1, dp2, None, dpx, None, None, None, None


def linear_wgrad(dy, x, gemm_dtype="bf16", want_bias=True):
    """dW [O,I] = dy^T x (+ db = column sums of dy) on the encoder GEMM (split over K = M rows when that pays)."""
    N.require_gpu(dy, x)
    dy, x = dy.contiguous(), x.contiguous()
    M, O = dy.shape
    I = x.shape[1]
    dt = N.BF16 if gemm_dtype == "bf16" else N.F32
    dw = torch.empty(O, I, dtype=torch.float32, device=x.device)
    db = torch.empty(O, dtype=torch.float32, device=x.device) if want_bias else None
    ws = N.workspace(N.lib().cmh_linear_wgrad_workspace_bytes(dt, M, O, I), x.device, "wgrad")
    N.check(N.lib().cmh_linear_wgrad(dt, N.ptr(dy), KIND[dy.dtype], N.ptr(x), KIND[x.dtype], M, O, I, N.ptr(dw), N.ptr(db), N.ptr(ws),
                                     ws.numel(), N.stream_ptr(x.device)), "cmh_linear_wgrad")
    return dw, db


class PairSoftmax(torch.autograd.Function):
    """cmh_pair_softmax with its backward (DCHMT select head, model/DCHMT.py:24)."""

    @staticmethod
    def forward(ctx, z):
        p = N.pair_softmax(z)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        dp = N.f32c(dp)
        dz = torch.empty_like(p)
        N.check(N.lib().cmh_pair_softmax_backward(N.ptr(p), N.ptr(dp), N.ptr(dz), p.shape[0], p.shape[1] // 2,
                                                  N.stream_ptr(p.device)), "cmh_pair_softmax_backward")
        return dz


class DchmtLoss(torch.autograd.Function):
    """cmh_dchmt_loss with its backward (similarity_loss x3 + our_loss, train/DCHMT/hash_train.py:82-150)."""

    @staticmethod
    def forward(ctx, img, txt, label, output_dim, similarity, loss_type, vartheta, sim_threshold):
        img, txt, label = N.f32c(img), N.f32c(txt), N.f32c(label)
        ctx.save_for_backward(img, txt, label)
        ctx.cfg = (int(output_dim), similarity, loss_type, float(vartheta), float(sim_threshold))
        return N.dchmt_loss(img, txt, label, output_dim, similarity, loss_type, vartheta, sim_threshold)

    @staticmethod
    def backward(ctx, dloss):
        img, txt, label = ctx.saved_tensors
        K, similarity, loss_type, vartheta, thr = ctx.cfg
        B, D = img.shape
        Cn = label.shape[1]
        di, dt = torch.empty_like(img), torch.empty_like(txt)
        dl = N.f32c(dloss).reshape(1)
        ws = N.workspace(N.lib().cmh_head_backward_workspace_bytes(B, D, Cn), img.device, "bwd")
        N.check(N.lib().cmh_dchmt_loss_backward(N.ptr(img), N.ptr(txt), N.ptr(label), B, D, Cn, K,
                                                {"euclidean": 0, "cosine": 1}[similarity], {"l1": 1, "l2": 2}[loss_type], vartheta, thr,
                                                N.ptr(dl), N.ptr(di), N.ptr(dt), N.ptr(ws), ws.numel(), N.stream_ptr(img.device)),
                "cmh_dchmt_loss_backward")
        return di, dt, None, None, None, None, None, None
