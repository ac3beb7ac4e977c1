"""Autograd bridges of the two towers: forward = cmh_*_forward_train (keeps a tape), backward = cmh_*_backward, which
writes one f32 gradient per parameter (include/cmh.h "Training forward ... and backward").  Used by CLIP.encode_image /
encode_text whenever gradients are enabled and a tower parameter requires them; under torch.no_grad() the plain encode
entry points run instead."""
import ctypes as C

import torch

import cmh_native as N

_BLOCK_FIELDS = ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln1_w", "ln1_b", "ln2_w", "ln2_b",
                 "fc_w", "fc_b", "proj_w", "proj_b")


def block_params(blk):
    """the 12 parameters of one ResidualAttentionBlock in cmh_block_weights order"""
    return [blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.attn.out_proj.weight, blk.attn.out_proj.bias,
            blk.ln_1.weight, blk.ln_1.bias, blk.ln_2.weight, blk.ln_2.bias,
            blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias]


def vit_params(v):
    head = [v.conv1.weight, v.class_embedding, v.positional_embedding, v.ln_pre.weight, v.ln_pre.bias,
            v.ln_post.weight, v.ln_post.bias, v.proj]
    return head + [p for blk in v.transformer.resblocks for p in block_params(blk)]


def text_params(clip):
    head = [clip.token_embedding.weight, clip.positional_embedding, clip.ln_final.weight, clip.ln_final.bias,
            clip.text_projection]
    return head + [p for blk in clip.transformer.resblocks for p in block_params(blk)]


# Data-parallel hook (dist_utils.GradSync): called as BUCKET_SINK(flat_slice, params, views) from inside a tower's backward the
# moment a part of the pass has written its last gradient, with the 1-D f32 slice of the flat buffer that holds exactly that part's
# gradients (views[i] = the gradient of params[i] inside it); the sink may queue an in-place all-reduce on the slice: the values
# reach the parameters' .grad (those very views, once autograd has adopted them) without any packing or copy-back.
BUCKET_SINK = None
PARTS = 3          # tower backward calls per step: blocks in PARTS near-equal ranges, last layers first


def _grad_buffers(params, order=None):
    """One flat f32 buffer for all gradients of a tower call; returns (grads as views in `params` order, flat).  `order` lists
    the parameter indices in the order the backward pass completes them, so that every part is one contiguous slice."""
    for p in params:
        if p.dtype != torch.float32 or not p.is_contiguous():
            raise N.NativeError("training needs contiguous float32 parameters (model.float())")
    order = list(range(len(params))) if order is None else order
    sizes = [(params[i].numel() + 3) // 4 * 4 for i in order]                 # 16-byte aligned views
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=params[0].device)
    grads, off = [None] * len(params), 0
    for i, n in zip(order, sizes):
        grads[i] = flat[off:off + params[i].numel()].view_as(params[i])
        off += n
    return grads, flat


def _layer_parts(layers, parts=None):
    """[(hi, lo), ...] block ranges of the backward calls, last layers first"""
    parts = max(1, min(parts or PARTS, layers)) if layers > 0 else 1
    cuts = [layers - (layers * k) // parts for k in range(parts + 1)]
    return [(cuts[k], cuts[k + 1]) for k in range(parts)]


def _part_order(nhead, head_first, layers, ranges):
    """Flat-buffer order and slice boundaries: [head parameters that the first part completes][blocks of part 0, last first] ...
    [blocks of the last part][head parameters that the last part completes].  -> (order, element boundaries per part)"""
    order, bounds = list(head_first), []
    for k, (hi, lo) in enumerate(ranges):
        for layer in range(hi - 1, lo - 1, -1):
            order += list(range(nhead + 12 * layer, nhead + 12 * (layer + 1)))
        if k == len(ranges) - 1:
            order += [i for i in range(nhead) if i not in head_first]
        bounds.append(len(order))
    return order, bounds


def _sink(flat, params, grads, order, bounds, k):
    if BUCKET_SINK is None:
        return
    size = lambda i: (params[i].numel() + 3) // 4 * 4
    lo = sum(size(i) for i in order[:bounds[k - 1]]) if k else 0
    hi = sum(size(i) for i in order[:bounds[k]])
    idx = order[(bounds[k - 1] if k else 0):bounds[k]]
    BUCKET_SINK(flat[lo:hi], [params[i] for i in idx], [grads[i] for i in idx])


def _block_grads(grads, nhead):
    layers = (len(grads) - nhead) // 12
    arr = (N.BlockGrads * layers)()
    for i in range(layers):
        for j, f in enumerate(_BLOCK_FIELDS):
            setattr(arr[i], f, grads[nhead + 12 * i + j].data_ptr())
    return arr


class VitTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, image, *params):
        s = clip._vit_struct()
        B = image.shape[0]
        feat = torch.empty(B, s.embed_dim, dtype=torch.float32, device=image.device)
        tape = torch.empty(N.lib().cmh_vit_train_bytes(C.byref(s), B), dtype=torch.uint8, device=image.device)
        N.check(N.lib().cmh_vit_forward_train(C.byref(s), N.ptr(image), B, N.ptr(feat), N.ptr(tape), tape.numel(),
                                              N.stream_ptr(image.device)), "cmh_vit_forward_train")
        ctx.clip, ctx.tape, ctx.B, ctx.struct = clip, tape, B, s
        ctx.params = params
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        s, params = ctx.struct, ctx.params
        ranges = _layer_parts(s.layers)
        order, bounds = _part_order(8, (5, 6, 7), s.layers, ranges)       # ln_post.weight / bias, proj come with the first part
        grads, flat = _grad_buffers(params, order)
        blocks = _block_grads(grads, 8)
        g = N.VitGrads(*[t.data_ptr() for t in grads[:8]], C.cast(blocks, C.POINTER(N.BlockGrads)))
        dfeat = N.f32c(dfeat)
        for k, (hi, lo) in enumerate(ranges):
            N.check(N.lib().cmh_vit_backward_part(C.byref(s), ctx.B, N.ptr(dfeat), C.byref(g), N.ptr(ctx.tape), ctx.tape.numel(), hi, lo,
                                                  N.stream_ptr(dfeat.device)), "cmh_vit_backward_part")
            _sink(flat, params, grads, order, bounds, k)
        ctx.tape = None
        return (None, None) + tuple(grads)


class TextTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, text, kpm, *params):
        s = clip._text_struct()
        B, L = text.shape
        feat = torch.empty(B, s.embed_dim, dtype=torch.float32, device=text.device)
        tape = torch.empty(N.lib().cmh_text_train_bytes(C.byref(s), B, L), dtype=torch.uint8, device=text.device)
        N.check(N.lib().cmh_text_forward_train(C.byref(s), N.ptr(text), B, L, N.ptr(kpm), N.ptr(feat), N.ptr(tape), tape.numel(),
                                               N.stream_ptr(text.device)), "cmh_text_forward_train")
        ctx.tape, ctx.struct, ctx.text, ctx.kpm, ctx.params = tape, s, text, kpm, params
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        s, params, text = ctx.struct, ctx.params, ctx.text
        ranges = _layer_parts(s.layers)
        order, bounds = _part_order(5, (2, 3, 4), s.layers, ranges)       # ln_final.weight / bias, text_projection: first part
        grads, flat = _grad_buffers(params, order)
        blocks = _block_grads(grads, 5)
        g = N.TextGrads(*[t.data_ptr() for t in grads[:5]], C.cast(blocks, C.POINTER(N.BlockGrads)))
        dfeat = N.f32c(dfeat)
        B, L = text.shape
        for k, (hi, lo) in enumerate(ranges):
            N.check(N.lib().cmh_text_backward_part(C.byref(s), N.ptr(text), B, L, N.ptr(ctx.kpm), N.ptr(dfeat), C.byref(g), N.ptr(ctx.tape),
                                                   ctx.tape.numel(), hi, lo, N.stream_ptr(dfeat.device)), "cmh_text_backward_part")
            _sink(flat, params, grads, order, bounds, k)
        ctx.tape = None
        return (None, None, None) + tuple(grads)


class VitTrainTokens(torch.autograd.Function):
    """The MITH trunk's image tower under training: every token projected (model/MITH.py:56-82) -> [B*T, E]."""

    @staticmethod
    def forward(ctx, clip, image, *params):
        s = clip._vit_struct()
        B = image.shape[0]
        T = (s.resolution // s.patch) ** 2 + 1
        tok = torch.empty(B * T, s.embed_dim, dtype=torch.float32, device=image.device)
        tape = torch.empty(N.lib().cmh_vit_train_bytes(C.byref(s), B), dtype=torch.uint8, device=image.device)
        N.check(N.lib().cmh_vit_forward_train_tokens(C.byref(s), N.ptr(image), B, N.ptr(tok), N.ptr(tape), tape.numel(),
                                                     N.stream_ptr(image.device)), "cmh_vit_forward_train_tokens")
        ctx.tape, ctx.B, ctx.struct, ctx.params = tape, B, s, params
        return tok

    @staticmethod
    def backward(ctx, dtok):
        s, params = ctx.struct, ctx.params
        grads, flat = _grad_buffers(params)
        blocks = _block_grads(grads, 8)
        g = N.VitGrads(*[t.data_ptr() for t in grads[:8]], C.cast(blocks, C.POINTER(N.BlockGrads)))
        dtok = N.f32c(dtok)
        N.check(N.lib().cmh_vit_backward_tokens(C.byref(s), ctx.B, N.ptr(dtok), C.byref(g), N.ptr(ctx.tape), ctx.tape.numel(),
                                                N.stream_ptr(dtok.device)), "cmh_vit_backward_tokens")
        if BUCKET_SINK is not None:
            BUCKET_SINK(flat, list(params), list(grads))
        ctx.tape = None
        return (None, None) + tuple(grads)


class TextTrainTokens(torch.autograd.Function):
    """The MITH trunk's text tower under training (model/MITH.py:120-144) -> ([B*L, E] tokens, eot rows i32 [B])."""

    @staticmethod
    def forward(ctx, clip, text, kpm, *params):
        s = clip._text_struct()
        B, L = text.shape
        tok = torch.empty(B * L, s.embed_dim, dtype=torch.float32, device=text.device)
        rows = torch.empty(B, dtype=torch.int32, device=text.device)
        tape = torch.empty(N.lib().cmh_text_train_bytes(C.byref(s), B, L), dtype=torch.uint8, device=text.device)
        # (padded_tokens_unused, set by MITH: nobody reads the padded positions - they are neither computed nor differentiated)
        packed = bool(getattr(clip, "padded_tokens_unused", False)) and kpm is not None
        fn = N.lib().cmh_text_forward_train_tokens_packed if packed else N.lib().cmh_text_forward_train_tokens
        N.check(fn(C.byref(s), N.ptr(text), B, L, N.ptr(kpm), N.ptr(tok), N.ptr(rows), N.ptr(tape), tape.numel(), N.stream_ptr(text.device)),
                "cmh_text_forward_train_tokens")
        ctx.tape, ctx.struct, ctx.text, ctx.kpm, ctx.params = tape, s, text, kpm, params
        ctx.mark_non_differentiable(rows)
        return tok, rows

    @staticmethod
    def backward(ctx, dtok, _drows):
        s, params, text = ctx.struct, ctx.params, ctx.text
        grads, flat = _grad_buffers(params)
        blocks = _block_grads(grads, 5)
        g = N.TextGrads(*[t.data_ptr() for t in grads[:5]], C.cast(blocks, C.POINTER(N.BlockGrads)))
        dtok = N.f32c(dtok)
        B, L = text.shape
        N.check(N.lib().cmh_text_backward_tokens(C.byref(s), N.ptr(text), B, L, N.ptr(ctx.kpm), N.ptr(dtok), C.byref(g),
                                                 N.ptr(ctx.tape), ctx.tape.numel(), N.stream_ptr(dtok.device)),
                "cmh_text_backward_tokens")
        if BUCKET_SINK is not None:
            BUCKET_SINK(flat, list(params), list(grads))
        ctx.tape = None
        return (None, None, None) + tuple(grads)


def wants_grad(params):
    return torch.is_grad_enabled() and any(p.requires_grad for p in params)
