"""CLIP ViT trunk with the reference's module tree and state_dict, executed by libcmh.so.

Mirror of the reference's model/base/model.py (ViT path only; the ModifiedResNet path is not
reachable with a ViT-B/32 checkpoint):
  LayerNorm / QuickGELU            model/base/model.py:153-164
  ResidualAttentionBlock           :167-196
  Transformer                      :199-207
  VisionTransformer                :210-252
  CLIP (encode_image/encode_text)  :255-372
  convert_weights / build_model    :391-455
The nn.Modules here are parameter containers with the SAME names and shapes (302 tensors for
ViT-B/32, so OpenAI `ViT-B-32.pt` and reference `model-N.pth` checkpoints load strict); `forward`
never touches ATen math: encode_image / encode_text call cmh_vit_encode / cmh_text_encode.

Arithmetic mode (`CLIP.set_gemm_dtype`): "f32" = exact fp32 MFMA, the parity mode for the
reference's `model.float()` trainers; "bf16" = bf16 MFMA operands with fp32 accumulation,
LayerNorm/softmax/residual in fp32 (the throughput mode, BASELINE.json configs[1]).
Backward through the towers is not implemented yet (SURVEY §8f "next" #2): outputs carry a grad_fn
that raises, so `loss.backward()` fails loudly instead of silently training nothing.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

import cmh_native as N


class LayerNorm(nn.LayerNorm):
    """Parameter container; statistics are always fp32 inside the HIP kernel."""


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x): fused into the c_fc GEMM epilogue."""


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, attn_mask=None):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution = input_resolution
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class _NoBackward(torch.autograd.Function):
    """Attaches a grad_fn to outputs of the inference-only entry points so that a backward pass through them fails loudly
    instead of silently stopping (the training entry points keep a tape and are used whenever gradients are wanted)."""

    @staticmethod
    def forward(ctx, out, anchor):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError(
            "this output came from an inference-only call (per-block taps, `assume_frozen`, or an operand that did not ask for "
            "gradients when it was produced): it carries no tape, so nothing can be propagated through it")


def no_backward(out: torch.Tensor, anchor: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and anchor.requires_grad:
        return _NoBackward.apply(out, anchor)
    return out


class _TowerCache:
    """Device-resident weights in the layout libcmh wants + the ctypes structs that point at them.
    Rebuilt when any parameter changed (data_ptr/_version) or the arithmetic mode changed."""

    def __init__(self):
        self.key = None
        self.keep = []          # tensors that must outlive the structs
        self.struct = None
        self.blocks = None


def _prep(p: torch.Tensor, keep: list) -> torch.Tensor:
    t = p.detach()
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    keep.append(t)
    return t


def _gemm_w(p: torch.Tensor, dt: int, keep: list, transpose: bool = False) -> torch.Tensor:
    if dt in (N.BF16, N.FP8) and not transpose and p.dtype == torch.float32 and p.is_contiguous():
        # the bf16 copy lives on the parameter: the fused BertAdam step rewrites it together with the f32 values (cmh_adam_tensor.
        # p_bf16) and stamps the version it is current for, so a training step pays no cast pass over the weights
        t = getattr(p, "_cmh_bf16", None)
        if t is None or t.device != p.device or t.numel() != p.numel() or getattr(p, "_cmh_bf16_version", None) != (p.data_ptr(), p._version):
            t = N.cast_bf16(p.detach().reshape(p.shape[0], -1))
            p._cmh_bf16, p._cmh_bf16_version = t, (p.data_ptr(), p._version)
        keep.append(t)
        return t
    t = p.detach().float()
    t = t.t().contiguous() if transpose else t.contiguous()
    t = t.reshape(t.shape[0], -1)
    if dt in (N.BF16, N.FP8):       # outside the blocks the fp8 mode is the bf16 mode (conv1, the final projections)
        t = N.cast_bf16(t)
    keep.append(t)
    return t


def _fp8_w(p: torch.Tensor, keep: list):
    """Linear weight [out, in] -> (e4m3 bytes, per-output-channel scales): the tensors convert_weights lowers upstream."""
    q, cs = N.fp8_quantize_weight(p.detach().float().contiguous())
    keep.extend((q, cs))
    return q.data_ptr(), cs.data_ptr()


FP8_HEADROOM = 2.0     # act_scale = FP8_HEADROOM * amax(calibration batch) / 448: later batches may exceed the calibration maximum


def _fill_blocks(resblocks, dt: int, keep: list, act_amax=None):
    arr = (N.BlockWeights * len(resblocks))()
    for i, blk in enumerate(resblocks):
        b = arr[i]
        if dt == N.FP8:
            if act_amax is None:
                raise N.NativeError("fp8 mode needs activation scales: call clip.calibrate_fp8(images, texts) first")
            b.in_proj_w, b.in_proj_cs = _fp8_w(blk.attn.in_proj_weight, keep)
            b.out_proj_w, b.out_proj_cs = _fp8_w(blk.attn.out_proj.weight, keep)
            b.fc_w, b.fc_cs = _fp8_w(blk.mlp.c_fc.weight, keep)
            b.proj_w, b.proj_cs = _fp8_w(blk.mlp.c_proj.weight, keep)
            for j in range(4):
                b.act_scale[j] = max(float(act_amax[i][j]), 1e-6) * FP8_HEADROOM / 448.0
            for name, src in (("in_proj_b", blk.attn.in_proj_bias), ("out_proj_b", blk.attn.out_proj.bias), ("ln1_w", blk.ln_1.weight),
                              ("ln1_b", blk.ln_1.bias), ("ln2_w", blk.ln_2.weight), ("ln2_b", blk.ln_2.bias),
                              ("fc_b", blk.mlp.c_fc.bias), ("proj_b", blk.mlp.c_proj.bias)):
                setattr(b, name, _prep(src, keep).data_ptr())
            continue
        b.in_proj_w = _gemm_w(blk.attn.in_proj_weight, dt, keep).data_ptr()
        b.in_proj_b = _prep(blk.attn.in_proj_bias, keep).data_ptr()
        b.out_proj_w = _gemm_w(blk.attn.out_proj.weight, dt, keep).data_ptr()
        b.out_proj_b = _prep(blk.attn.out_proj.bias, keep).data_ptr()
        b.ln1_w = _prep(blk.ln_1.weight, keep).data_ptr()
        b.ln1_b = _prep(blk.ln_1.bias, keep).data_ptr()
        b.ln2_w = _prep(blk.ln_2.weight, keep).data_ptr()
        b.ln2_b = _prep(blk.ln_2.bias, keep).data_ptr()
        b.fc_w = _gemm_w(blk.mlp.c_fc.weight, dt, keep).data_ptr()
        b.fc_b = _prep(blk.mlp.c_fc.bias, keep).data_ptr()
        b.proj_w = _gemm_w(blk.mlp.c_proj.weight, dt, keep).data_ptr()
        b.proj_b = _prep(blk.mlp.c_proj.bias, keep).data_ptr()
    return arr


_DT = {"f32": N.F32, "fp32": N.F32, "float32": N.F32, "bf16": N.BF16, "bfloat16": N.BF16, "fp8": N.FP8, "e4m3": N.FP8}


class CLIP(nn.Module):
    def __init__(self, embed_dim: int, image_resolution: int, vision_layers, vision_width: int,
                 vision_patch_size: int, context_length: int, vocab_size: int, transformer_width: int,
                 transformer_heads: int, transformer_layers: int):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet (RN50) visual towers are out of scope; ViT checkpoints only")
        self.context_length = context_length
        self.vision_heads = vision_width // 64
        self.visual = VisionTransformer(input_resolution=image_resolution, patch_size=vision_patch_size,
                                        width=vision_width, layers=vision_layers, heads=self.vision_heads,
                                        output_dim=embed_dim)
        self.transformer = Transformer(width=transformer_width, layers=transformer_layers, heads=transformer_heads,
                                       attn_mask=self.build_attention_mask)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.initialize_parameters()
        self._gemm_dtype = _DT[os.environ.get("CMH_GEMM_DTYPE", "f32").lower()]
        self._vit_cache = _TowerCache()
        self._txt_cache = _TowerCache()
        self.assume_frozen = False     # True: skip the per-call parameter-version scan (pure inference)
        self._fp8_amax = {"vit": None, "text": None}     # [layers, 4] calibration maxima of the four GEMM inputs per block
        self._fp8_amax_key = {"vit": None, "text": None}  # the parameter versions those maxima were measured with

    # -- reference API ---------------------------------------------------------------------------
    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        for tr in (self.transformer,):
            proj_std = (tr.width ** -0.5) * ((2 * tr.layers) ** -0.5)
            attn_std = tr.width ** -0.5
            fc_std = (2 * tr.width) ** -0.5
            for block in tr.resblocks:
                nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
                nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
                nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
                nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)

    def build_attention_mask(self, context_length):
        """Kept for API parity; the kernel applies the causal mask itself."""
        mask = torch.empty(context_length, context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask.to(self.device)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    @property
    def device(self):
        return self.visual.conv1.weight.device

    def set_gemm_dtype(self, name: str):
        """"f32" (parity mode), "bf16" (throughput mode) or "fp8" (BASELINE configs[4]: the blocks' four GEMMs on e4m3 operands;
        inference only; activation scales come from `calibrate_fp8`, run on the first batch if it was not called)."""
        self._gemm_dtype = _DT[name.lower()]
        return self

    def calibrate_fp8(self, image=None, text=None, reset=True):
        """Calibration pass of the fp8 mode: runs the batch(es) in bf16 mode and records, per block, the largest magnitude of the
        four GEMM inputs (ln_1 output, attention output, ln_2 output, QuickGELU(c_fc) output).  The per-tensor quantisation
        scales are FP8_HEADROOM * amax / 448.  Call again with reset=False to widen the maxima with more batches.
        Returns the bf16-mode features of the batches it was given."""
        mode, self._gemm_dtype = self._gemm_dtype, N.BF16
        out = []
        try:
            with torch.no_grad():
                for tower, x in (("vit", image), ("text", text)):
                    if x is None:
                        continue
                    s = self._vit_struct() if tower == "vit" else self._text_struct()
                    dev = x.device
                    N.require_gpu(x)
                    prev = self._fp8_amax[tower]
                    amax = torch.zeros(s.layers * 4, dtype=torch.float32, device=dev)
                    if prev is not None and not reset:
                        amax.copy_(torch.as_tensor(prev, dtype=torch.float32).reshape(-1))
                    B = x.shape[0]
                    feat = torch.empty(B, s.embed_dim, dtype=torch.float32, device=dev)
                    if tower == "vit":
                        x = N.f32c(x)
                        ws = N.workspace(N.lib().cmh_vit_workspace_bytes(C.byref(s), B), dev, f"vit@{N.stream_ptr(dev)}")
                        N.check(N.lib().cmh_vit_calibrate_fp8(C.byref(s), N.ptr(x), B, N.ptr(feat), N.ptr(amax), N.ptr(ws), ws.numel(),
                                                              N.stream_ptr(dev)), "cmh_vit_calibrate_fp8")
                    else:
                        x = x.to(torch.int64).contiguous()
                        L = x.shape[1]
                        ws = N.workspace(N.lib().cmh_text_workspace_bytes(C.byref(s), B, L), dev, f"text@{N.stream_ptr(dev)}")
                        N.check(N.lib().cmh_text_calibrate_fp8(C.byref(s), N.ptr(x), B, L, N.ptr(feat), N.ptr(amax), N.ptr(ws), ws.numel(),
                                                               N.stream_ptr(dev)), "cmh_text_calibrate_fp8")
                    vals = amax.cpu().reshape(s.layers, 4)
                    if not bool(torch.isfinite(vals).all()):
                        raise N.NativeError("calibrate_fp8: non-finite activations in the calibration batch")
                    self._fp8_amax[tower] = vals.tolist()
                    self._fp8_amax_key[tower] = self._fp8_param_key(tower)
                    (self._vit_cache if tower == "vit" else self._txt_cache).key = None       # scales changed: rebuild the fp8 structs
                    out.append(feat)
        finally:
            self._gemm_dtype = mode
        return tuple(out)

    # encode_text skips the padding after each caption's EOT (bit-identical pooled features, cmh_text_encode_packed);
    # CMH_TEXT_PACK=0 or `clip.pack_text = False` computes all context_length positions like the reference
    pack_text = os.environ.get("CMH_TEXT_PACK", "1") != "0"
    _last_text_rows = None

    @property
    def last_text_rows(self):
        """(rows computed, batch * seq_len) of the last packed encode_text, or None; reading it fetches the count from the device
        (the only synchronisation: the encode itself never waits for it)."""
        if self._last_text_rows is None:
            return None
        rows, dense = self._last_text_rows
        return int(rows.item()), dense

    @property
    def gemm_dtype(self) -> str:
        return {N.BF16: "bf16", N.FP8: "fp8"}.get(self._gemm_dtype, "f32")

    def _fp8_param_key(self, tower):
        blocks = self.visual.transformer if tower == "vit" else self.transformer
        return tuple((p.data_ptr(), p._version) for p in blocks.parameters())

    def _fp8_scales_current(self, tower):
        """Activation scales belong to the weights they were calibrated with: after load_state_dict or a training step the blocks
        produce other magnitudes, and stale per-tensor scales saturate at +-448 or waste range.  A change of any block parameter
        since the calibration drops the maxima, so that the next fp8 batch calibrates itself (assume_frozen skips the check)."""
        if self._fp8_amax[tower] is not None and not self.assume_frozen and self._fp8_amax_key[tower] != self._fp8_param_key(tower):
            import warnings
            warnings.warn(f"fp8 mode: the {tower} tower's weights changed since calibrate_fp8; re-calibrating on this batch")
            self._fp8_amax[tower] = None
        return self._fp8_amax[tower] is not None

    # -- native weight structs -------------------------------------------------------------------
    def _key(self, params):
        if self.assume_frozen:
            return (self._gemm_dtype, str(self.device), "frozen")
        return (self._gemm_dtype, str(self.device)) + tuple((p.data_ptr(), p._version) for p in params)

    def _vit_struct(self):
        v, c = self.visual, self._vit_cache
        key = self._key(list(v.parameters()))
        if c.key != key:
            keep, dt = [], self._gemm_dtype
            s = N.VitWeights()
            s.gemm_dtype = dt
            s.resolution = v.input_resolution
            s.patch = v.conv1.weight.shape[-1]
            s.width = v.conv1.weight.shape[0]
            s.layers = len(v.transformer.resblocks)
            s.embed_dim = v.proj.shape[1]
            s.conv1_w = _gemm_w(v.conv1.weight, dt, keep).data_ptr()
            s.class_embedding = _prep(v.class_embedding, keep).data_ptr()
            s.positional_embedding = _prep(v.positional_embedding, keep).data_ptr()
            s.ln_pre_w = _prep(v.ln_pre.weight, keep).data_ptr()
            s.ln_pre_b = _prep(v.ln_pre.bias, keep).data_ptr()
            s.ln_post_w = _prep(v.ln_post.weight, keep).data_ptr()
            s.ln_post_b = _prep(v.ln_post.bias, keep).data_ptr()
            s.proj_t = _gemm_w(v.proj, dt, keep, transpose=True).data_ptr()
            c.blocks = _fill_blocks(v.transformer.resblocks, dt, keep, self._fp8_amax["vit"])
            s.blocks = C.cast(c.blocks, C.POINTER(N.BlockWeights))
            c.struct, c.keep, c.key = s, keep, key
        return c.struct

    def _text_struct(self):
        c = self._txt_cache
        params = [self.token_embedding.weight, self.positional_embedding, self.ln_final.weight, self.ln_final.bias,
                  self.text_projection] + list(self.transformer.parameters())
        key = self._key(params)
        if c.key != key:
            keep, dt = [], self._gemm_dtype
            s = N.TextWeights()
            s.gemm_dtype = dt
            s.context_length = self.positional_embedding.shape[0]
            s.vocab_size = self.token_embedding.weight.shape[0]
            s.width = self.ln_final.weight.shape[0]
            s.layers = len(self.transformer.resblocks)
            s.embed_dim = self.text_projection.shape[1]
            s.token_embedding = _prep(self.token_embedding.weight, keep).data_ptr()
            s.positional_embedding = _prep(self.positional_embedding, keep).data_ptr()
            s.ln_final_w = _prep(self.ln_final.weight, keep).data_ptr()
            s.ln_final_b = _prep(self.ln_final.bias, keep).data_ptr()
            s.text_projection_t = _gemm_w(self.text_projection, dt, keep, transpose=True).data_ptr()
            c.blocks = _fill_blocks(self.transformer.resblocks, dt, keep, self._fp8_amax["text"])
            s.blocks = C.cast(c.blocks, C.POINTER(N.BlockWeights))
            c.struct, c.keep, c.key = s, keep, key
        return c.struct

    @staticmethod
    def _taps(tap_list):
        if not tap_list:
            return None, None
        arr = (C.c_void_p * len(tap_list))(*[None if t is None else t.data_ptr() for t in tap_list])
        taps = N.Taps(C.cast(arr, C.POINTER(C.c_void_p)), len(tap_list))
        return taps, arr

    # -- the two hot-path entry points -----------------------------------------------------------
    def encode_image(self, image, taps=None):
        """image f32 [B,3,R,R] -> [B, embed_dim] f32  (reference model/base/model.py:356-357,228-252)."""
        if taps is None:
            hit = self._stashed("image", image)
            if hit is not None:
                return hit
        image = N.f32c(image)
        N.require_gpu(image, self.visual.proj)
        if self._gemm_dtype == N.FP8 and not self._fp8_scales_current("vit"):
            self.calibrate_fp8(image=image)              # first batch (or first after a weight change) calibrates the activation scales
        s = self._vit_struct()
        B = image.shape[0]
        if tuple(image.shape[1:]) != (3, s.resolution, s.resolution):
            raise N.NativeError(f"encode_image: expected [B,3,{s.resolution},{s.resolution}], got {tuple(image.shape)}")
        if taps is None and not self.assume_frozen:
            from model.base import train_ops as T            # training: tape-keeping forward with a real backward
            params = T.vit_params(self.visual)
            if T.wants_grad(params):
                if self._gemm_dtype == N.FP8:
                    raise N.NativeError("the fp8 mode is inference-only: use torch.no_grad() or set_gemm_dtype('bf16' | 'f32') to train")
                return T.VitTrain.apply(self, image, *params)
        s = self._vit_struct()
        feat = torch.empty(B, s.embed_dim, dtype=torch.float32, device=image.device)
        need = N.lib().cmh_vit_workspace_bytes(C.byref(s), B)
        ws = N.workspace(need, image.device, f"vit@{N.stream_ptr(image.device)}")   # one scratch per stream: sub-batches may overlap
        tp, _arr = self._taps(taps)
        N.check(N.lib().cmh_vit_encode(C.byref(s), N.ptr(image), B, N.ptr(feat), N.ptr(ws), ws.numel(),
                                       None if tp is None else C.byref(tp), N.stream_ptr(image.device)),
                "cmh_vit_encode")
        return no_backward(feat, self.visual.proj)

    def encode_text(self, text, key_padding_mask=None, taps=None):
        """text i64 [B,L] -> [B, embed_dim] f32  (reference model/base/model.py:359-372)."""
        if taps is None and key_padding_mask is None:
            hit = self._stashed("text", text)
            if hit is not None:
                return hit
        N.require_gpu(text, self.text_projection)
        text = text.to(torch.int64).contiguous()
        if self._gemm_dtype == N.FP8 and not self._fp8_scales_current("text"):
            self.calibrate_fp8(text=text)
        s = self._text_struct()
        B, L = text.shape
        kpm = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
        if taps is None and not self.assume_frozen:
            from model.base import train_ops as T
            params = T.text_params(self)
            if T.wants_grad(params):
                if self._gemm_dtype == N.FP8:
                    raise N.NativeError("the fp8 mode is inference-only: use torch.no_grad() or set_gemm_dtype('bf16' | 'f32') to train")
                return T.TextTrain.apply(self, text, kpm, *params)
        s = self._text_struct()
        feat = torch.empty(B, s.embed_dim, dtype=torch.float32, device=text.device)
        need = N.lib().cmh_text_workspace_bytes(C.byref(s), B, L)
        ws = N.workspace(need, text.device, f"text@{N.stream_ptr(text.device)}")
        if kpm is None and taps is None and self.pack_text:
            # only the tokens up to each caption's EOT can reach the pooled row under the causal mask: skip the padding
            rows = torch.empty(1, dtype=torch.int32, device=text.device)       # the count stays on the device: nothing waits for it
            N.check(N.lib().cmh_text_encode_packed(C.byref(s), N.ptr(text), B, L, N.ptr(feat), N.ptr(rows), N.ptr(ws), ws.numel(),
                                                   N.stream_ptr(text.device)), "cmh_text_encode_packed")
            self._last_text_rows = (rows, B * L)
            return no_backward(feat, self.text_projection)
        tp, _arr = self._taps(taps)
        N.check(N.lib().cmh_text_encode(C.byref(s), N.ptr(text), B, L, N.ptr(kpm), N.ptr(feat), N.ptr(ws), ws.numel(),
                                        None if tp is None else C.byref(tp), N.stream_ptr(text.device)),
                "cmh_text_encode")
        return no_backward(feat, self.text_projection)

    def prefetch_pair(self, image, text):
        """Inference helper for code that calls encode_image(image) and encode_text(text) back to back (every method model's
        encode_* does, model/modelbase.py:88-92): run both towers NOW in lock-step (encode_pair) and hand the two features out to
        the next encode_image / encode_text call on these very tensors.  Bit-identical to the separate calls; a no-op when a
        gradient is wanted."""
        if torch.is_grad_enabled() and not self.assume_frozen:
            return
        fi, ft = self.encode_pair(image, text)
        key = self._stash_key()
        self._pair_stash = {"image": [(image, fi, key)], "text": [(text, ft, key)]}

    def prefetch_pairs(self, batches):
        """prefetch_pair for TWO loader batches [(image_a, text_a), (image_b, text_b)] in one native call (cmh_clip_encode_pair2: the
        batches' rows share every launch; a launch's fixed costs are paid once per 2 x 256 pairs).  Each batch's features are handed
        out to its own encode_image / encode_text calls; bit-identical to two prefetch_pair calls."""
        if torch.is_grad_enabled() and not self.assume_frozen:
            return
        (ia, ta), (ib, tb) = batches
        fi, ft = self.encode_pair2(ia, ta, ib, tb)
        key, Ba = self._stash_key(), ia.shape[0]
        self._pair_stash = {"image": [(ia, fi[:Ba], key), (ib, fi[Ba:], key)], "text": [(ta, ft[:Ba], key), (tb, ft[Ba:], key)]}

    def _stashed(self, side, x):
        st = getattr(self, "_pair_stash", None)
        if not st or not st.get(side):
            return None
        for i, (src, feat, key) in enumerate(st[side]):
            if src is x and key == self._stash_key():
                del st[side][i]
                if not st[side]:
                    del st[side]
                return feat
        return None

    def _stash_key(self):
        """what a stashed feature depends on besides its input tensor: the arithmetic mode and the weights' versions"""
        text = [self.token_embedding.weight, self.positional_embedding, self.ln_final.weight, self.ln_final.bias,
                self.text_projection] + list(self.transformer.parameters())
        return (self.pack_text, self._key(list(self.visual.parameters())), self._key(text))

    def drop_pair_stash(self):
        """Forget features prefetch_pair computed and nobody picked up; counts them (`pair_stash_misses`) so that a model whose
        encode_* calls never hit the stash - a mask, taps, other tensors - shows up instead of silently paying for both paths."""
        st = getattr(self, "_pair_stash", None)
        if st:
            self.pair_stash_misses = getattr(self, "pair_stash_misses", 0) + sum(len(v) for v in st.values())
        self._pair_stash = {}

    def encode_pair(self, image, text):
        """(encode_image(image), encode_text(text)) with the two towers in lock-step: layer i of both shares its GEMM launches
        (cmh_clip_encode_pair; reference model/modelbase.py:105-108 calls the two encoders back to back).  Bit-identical to the two
        separate calls.  Inference (no tape); training, taps and masks take the single-tower entry points."""
        if not self.assume_frozen and torch.is_grad_enabled():
            from model.base import train_ops as T
            if T.wants_grad(T.vit_params(self.visual)) or T.wants_grad(T.text_params(self)):
                return self.encode_image(image), self.encode_text(text)
        image = N.f32c(image)
        N.require_gpu(image, text, self.visual.proj, self.text_projection)
        text = text.to(torch.int64).contiguous()
        if self._gemm_dtype == N.FP8 and not (self._fp8_scales_current("vit") and self._fp8_scales_current("text")):
            self.calibrate_fp8(image=image, text=text)
        sv, st = self._vit_struct(), self._text_struct()
        B, L = text.shape
        if tuple(image.shape) != (B, 3, sv.resolution, sv.resolution):
            raise N.NativeError(f"encode_pair: expected image [{B},3,{sv.resolution},{sv.resolution}], got {tuple(image.shape)}")
        fi = torch.empty(B, sv.embed_dim, dtype=torch.float32, device=image.device)
        ft = torch.empty(B, st.embed_dim, dtype=torch.float32, device=image.device)
        tag = N.stream_ptr(image.device)
        wv = N.workspace(N.lib().cmh_vit_workspace_bytes(C.byref(sv), B), image.device, f"vit@{tag}")
        wt = N.workspace(N.lib().cmh_text_workspace_bytes(C.byref(st), B, L), image.device, f"text@{tag}")
        rows = torch.empty(1, dtype=torch.int32, device=image.device) if self.pack_text else None
        N.check(N.lib().cmh_clip_encode_pair(C.byref(sv), N.ptr(image), C.byref(st), N.ptr(text), B, L, 1 if self.pack_text else 0,
                                             N.ptr(fi), N.ptr(ft), N.ptr(rows), N.ptr(wv), wv.numel(), N.ptr(wt), wt.numel(), tag),
                "cmh_clip_encode_pair")
        if rows is not None:
            self._last_text_rows = (rows, B * L)
        return no_backward(fi, self.visual.proj), no_backward(ft, self.text_projection)

    def encode_pair2(self, image_a, text_a, image_b, text_b):
        """encode_pair of two batches in one native call: (features of [a; b] images, of [a; b] captions).  The images stay two
        tensors (cmh_clip_encode_pair2 patchifies them into one matrix), the captions are concatenated (157 KB).  Inference only."""
        image_a, image_b = N.f32c(image_a), N.f32c(image_b)
        N.require_gpu(image_a, image_b, text_a, text_b, self.visual.proj, self.text_projection)
        text = torch.cat([text_a.to(torch.int64), text_b.to(torch.int64)], 0).contiguous()
        if self._gemm_dtype == N.FP8 and not (self._fp8_scales_current("vit") and self._fp8_scales_current("text")):
            self.calibrate_fp8(image=image_a, text=text_a.to(torch.int64).contiguous())
        sv, st = self._vit_struct(), self._text_struct()
        Ba, Bb, L = image_a.shape[0], image_b.shape[0], text.shape[1]
        B = Ba + Bb
        for im, n in ((image_a, text_a.shape[0]), (image_b, text_b.shape[0])):
            if tuple(im.shape) != (n, 3, sv.resolution, sv.resolution):
                raise N.NativeError(f"encode_pair2: expected images [{n},3,{sv.resolution},{sv.resolution}], got {tuple(im.shape)}")
        fi = torch.empty(B, sv.embed_dim, dtype=torch.float32, device=image_a.device)
        ft = torch.empty(B, st.embed_dim, dtype=torch.float32, device=image_a.device)
        tag = N.stream_ptr(image_a.device)
        wv = N.workspace(N.lib().cmh_vit_workspace_bytes(C.byref(sv), B), image_a.device, f"vit@{tag}")
        wt = N.workspace(N.lib().cmh_text_workspace_bytes(C.byref(st), B, L), image_a.device, f"text@{tag}")
        rows = torch.empty(1, dtype=torch.int32, device=image_a.device) if self.pack_text else None
        N.check(N.lib().cmh_clip_encode_pair2(C.byref(sv), N.ptr(image_a), Ba, N.ptr(image_b), Bb, C.byref(st), N.ptr(text), L,
                                              1 if self.pack_text else 0, N.ptr(fi), N.ptr(ft), N.ptr(rows), N.ptr(wv), wv.numel(),
                                              N.ptr(wt), wt.numel(), tag), "cmh_clip_encode_pair2")
        if rows is not None:
            self._last_text_rows = (rows, B * L)
        return no_backward(fi, self.visual.proj), no_backward(ft, self.text_projection)

    def forward(self, image, text):
        """Cosine-similarity logits (reference :374-388); assembled from the two native encodes."""
        i = self.encode_image(image)
        t = self.encode_text(text)
        i = i / i.norm(dim=-1, keepdim=True)
        t = t / t.norm(dim=-1, keepdim=True)
        logits = self.logit_scale.exp() * i @ t.t()
        return logits, logits.t()


def convert_weights(model: nn.Module):
    """Reference :391-412: Conv/Linear weight+bias, MHA in_proj/bias and the two projections go to fp16.
    Checkpoints loaded afterwards are therefore rounded through fp16, exactly as upstream."""

    def _convert(l):
        if isinstance(l, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            l.weight.data = l.weight.data.half()
            if l.bias is not None:
                l.bias.data = l.bias.data.half()
        if isinstance(l, nn.MultiheadAttention):
            for attr in [*[f"{s}_proj_weight" for s in ["in", "q", "k", "v"]], "in_proj_bias", "bias_k", "bias_v"]:
                tensor = getattr(l, attr)
                if tensor is not None:
                    tensor.data = tensor.data.half()
        for name in ["text_projection", "proj"]:
            if hasattr(l, name):
                attr = getattr(l, name)
                if attr is not None:
                    attr.data = attr.data.half()

    model.apply(_convert)


def build_model(state_dict: dict):
    """Reference :415-455: infer the architecture from tensor shapes, fp16-convert, load strict."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("only ViT CLIP checkpoints are supported (no visual.proj in state_dict)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers)
    for key in ["input_resolution", "context_length", "vocab_size"]:
        if key in state_dict:
            del state_dict[key]
    convert_weights(model)
    model.load_state_dict(state_dict)
    return model
