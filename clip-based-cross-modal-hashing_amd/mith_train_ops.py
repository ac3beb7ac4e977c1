"""Autograd bridges of MITH's HashingModel (reference model/MITH.py:217-453): each Function is a libcmh forward paired with a
libcmh backward.  What upstream detaches stays detached here: the token-level concept similarities that steer the
aggregation (`gcl(tokens)[1].detach()`, :441-442 and :345), so LocalizedTokenAggregation is linear in the tokens."""
import ctypes as C

import torch

import backward_ops as B
import cmh_native as N
import mith_ops as M
from model.base.train_ops import _BLOCK_FIELDS, block_params


class GemmLinear(torch.autograd.Function):
    """y = x @ w.T + b on the encoder GEMM (f32 or bf16 operands); dx, dw, db on the same kernels (dgrad on w^T, split-K wgrad)."""

    @staticmethod
    def forward(ctx, x, w, b, dtype):
        x = N.f32c(x)
        ctx.save_for_backward(x, w)
        ctx.dtype, ctx.has_bias = dtype, b is not None
        return M.gemm(x, w, b, dtype=dtype)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = N.f32c(dy)
        mode = "bf16" if ctx.dtype == N.BF16 else "f32"
        if ctx.dtype == N.BF16:                                       # dx = dy @ (w^T)^T with w^T [I, O] as the GEMM's weight operand
            dx = N.linear_gemm(N.cast_bf16(dy), B.transpose(N.f32c(w), torch.bfloat16))
        else:
            dx = N.linear_gemm(dy, B.transpose(N.f32c(w)))
        dw, db = B.linear_wgrad(dy, x, gemm_dtype=mode, want_bias=ctx.has_bias)
        return dx, dw, db, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = N.f32c(x)
        ctx.save_for_backward(x)
        y = torch.empty_like(x)
        N.check(N.lib().cmh_gelu(N.ptr(x), N.ptr(y), x.numel(), N.stream_ptr(x.device)), "cmh_gelu")
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = N.f32c(dy)
        dx = torch.empty_like(x)
        N.check(N.lib().cmh_gelu_backward(N.ptr(x), N.ptr(dy), N.ptr(dx), x.numel(), N.stream_ptr(x.device)), "cmh_gelu_backward")
        return dx


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) over the rows of x [R, D]."""

    @staticmethod
    def forward(ctx, x):
        x = N.f32c(x)
        ctx.save_for_backward(x)
        return M.l2_normalize_rows(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = N.f32c(dy)
        dx = torch.empty_like(x)
        N.check(N.lib().cmh_l2_normalize_backward(N.ptr(x), N.ptr(dy), N.ptr(dx), x.shape[0], x.shape[1], N.stream_ptr(x.device)),
                "cmh_l2_normalize_backward")
        return dx


class LtaFn(torch.autograd.Function):
    """cmh_mith_lta: tokens [B, Ltot, D] aggregated into K concept tokens [B, K, D] with weights from the (detached) similarities."""

    @staticmethod
    def forward(ctx, tokens, sim, kpm, l0, L, top_k):
        tokens, sim = N.f32c(tokens), N.f32c(sim)
        kpm8 = None if kpm is None else kpm.to(torch.uint8).contiguous()
        ctx.save_for_backward(sim, kpm8)
        ctx.geom = (tokens.shape, l0, L, top_k)
        return M.lta(tokens, sim, kpm, l0, L, top_k)

    @staticmethod
    def backward(ctx, dmerge):
        sim, kpm8 = ctx.saved_tensors
        (Bn, Ltot, D), l0, L, top_k = ctx.geom
        dmerge = N.f32c(dmerge)
        dtok = torch.empty(Bn, Ltot, D, dtype=torch.float32, device=dmerge.device)
        N.check(N.lib().cmh_mith_lta_backward(N.ptr(sim), N.ptr(kpm8), N.ptr(dmerge), N.ptr(dtok), Bn, Ltot, l0, L, sim.shape[2], D, top_k,
                                              N.stream_ptr(dmerge.device)), "cmh_mith_lta_backward")
        return dtok, None, None, None, None, None


class AddPosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe):
        return M.add_positional(x, pe)

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class BitHashFn(torch.autograd.Function):
    """cmh_bitwise_hash: x [B, K, D], w [K, D], b [K] -> tanh codes [B, K]."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = N.f32c(x), N.f32c(w), N.f32c(b)
        y = M.bitwise_hash(x, w, b)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = N.f32c(dy)
        Bn, K, D = x.shape
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty(K, dtype=torch.float32, device=x.device)
        N.check(N.lib().cmh_bitwise_hash_backward(N.ptr(x), N.ptr(w), N.ptr(y), N.ptr(dy), N.ptr(dx), N.ptr(dw), N.ptr(db), Bn, K, D,
                                                  N.stream_ptr(x.device)), "cmh_bitwise_hash_backward")
        return dx, dw, db


class BlocksTrain(torch.autograd.Function):
    """A stack of ResidualAttentionBlocks over x f32 [B*T, d] with a tape (the concept transformer)."""

    @staticmethod
    def forward(ctx, x, Bn, T, dtype, block_array, keep, *params):
        x = N.f32c(x)
        layers = len(params) // 12
        d = x.shape[1]
        y = torch.empty_like(x)
        tape = torch.empty(N.lib().cmh_blocks_train_bytes(dtype, Bn, T, d, layers), dtype=torch.uint8, device=x.device)
        N.check(N.lib().cmh_blocks_forward_train(C.cast(block_array, C.POINTER(N.BlockWeights)), layers, dtype, N.ptr(x), N.ptr(y), Bn, T, d,
                                                 N.ptr(tape), tape.numel(), N.stream_ptr(x.device)), "cmh_blocks_forward_train")
        ctx.tape, ctx.geom, ctx.blocks, ctx.keep, ctx.params = tape, (Bn, T, d, dtype, layers), block_array, keep, params
        return y

    @staticmethod
    def backward(ctx, dy):
        Bn, T, d, dtype, layers = ctx.geom
        dy = N.f32c(dy)
        for p in ctx.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise N.NativeError("training needs contiguous float32 parameters (model.float())")
        grads = [torch.empty_like(p) for p in ctx.params]
        arr = (N.BlockGrads * layers)()
        for i in range(layers):
            for j, f in enumerate(_BLOCK_FIELDS):
                setattr(arr[i], f, grads[12 * i + j].data_ptr())
        dx = torch.empty_like(dy)
        N.check(N.lib().cmh_blocks_backward(C.cast(ctx.blocks, C.POINTER(N.BlockWeights)), C.cast(arr, C.POINTER(N.BlockGrads)), layers,
                                            dtype, N.ptr(dy), N.ptr(dx), Bn, T, d, N.ptr(ctx.tape), ctx.tape.numel(),
                                            N.stream_ptr(dy.device)), "cmh_blocks_backward")
        ctx.tape = ctx.keep = None
        return (dx, None, None, None, None, None) + tuple(grads)


def blocks_params(resblocks):
    return [p for blk in resblocks for p in block_params(blk)]


# ---- loss terms (train/MITH/hash_train.py:103-147) ----------------------------------------------------------------------------
class BayesianLossFn(torch.autograd.Function):
    """bayesian_loss(bank, batch, label_sim): the bank holds detached codes, so only the batch codes receive a gradient."""

    @staticmethod
    def forward(ctx, bank, batch, bank_label, label):
        bank, batch, bank_label, label = (N.f32c(t) for t in (bank, batch, bank_label, label))
        ctx.save_for_backward(bank.clone(), batch, bank_label, label)       # the trainer rewrites rows of the bank every step
        return M.bayesian_loss(bank, batch, bank_label, label).clone()

    @staticmethod
    def backward(ctx, g):
        bank, batch, bank_label, label = ctx.saved_tensors
        Mb, K = bank.shape
        Bn, Cn = label.shape
        db = torch.empty_like(batch)
        g = N.f32c(g).reshape(1)
        ws = N.workspace(N.lib().cmh_mith_bayesian_backward_workspace_bytes(Bn, K), bank.device, f"bayes@{N.stream_ptr(bank.device)}")
        N.check(N.lib().cmh_mith_bayesian_loss_backward(N.ptr(bank), N.ptr(batch), N.ptr(bank_label), N.ptr(label), Mb, Bn, K, Cn, N.ptr(g),
                                                        N.ptr(db), N.ptr(ws), ws.numel(), N.stream_ptr(bank.device)),
                "cmh_mith_bayesian_loss_backward")
        return None, db, None, None


class InfoNceFn(torch.autograd.Function):
    """Symmetric InfoNCE with diagonal targets inside groups of `group` consecutive rows (info_nce_loss / info_nce_loss_bmm)."""

    @staticmethod
    def forward(ctx, a, b, group, temperature):
        a, b = N.f32c(a), N.f32c(b)
        ctx.save_for_backward(a, b)
        ctx.group, ctx.temperature = group, float(temperature)
        return M.info_nce(a, b, group, temperature).clone()

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        D = a.shape[-1]
        R = a.numel() // D
        G = R if ctx.group is None else ctx.group
        da, db = torch.empty_like(a), torch.empty_like(b)
        g = N.f32c(g).reshape(1)
        ws = N.workspace(N.lib().cmh_info_nce_workspace_bytes(R, G), a.device, f"nce@{N.stream_ptr(a.device)}")
        N.check(N.lib().cmh_info_nce_backward(N.ptr(a), N.ptr(b), R, G, D, ctx.temperature, N.ptr(g), N.ptr(da), N.ptr(db), N.ptr(ws),
                                              ws.numel(), N.stream_ptr(a.device)), "cmh_info_nce_backward")
        return da, db, None, None


class SqDiffSumFn(torch.autograd.Function):
    """F.mse_loss(a, b, reduction='sum')."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = N.f32c(a), N.f32c(b)
        ctx.save_for_backward(a, b)
        return M.sq_diff_sum(a, b).clone()

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(b) if need_b else None
        g = N.f32c(g).reshape(1)
        N.check(N.lib().cmh_sq_diff_sum_backward(N.ptr(a), N.ptr(b), a.numel(), N.ptr(g), N.ptr(da), N.ptr(db), N.stream_ptr(a.device)),
                "cmh_sq_diff_sum_backward")
        return da, db
