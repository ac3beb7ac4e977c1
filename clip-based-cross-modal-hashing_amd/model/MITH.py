"""MITH model (reference model/MITH.py) on libcmh: token-returning CLIP trunk (ViT / CLIP1, :49-160) and HashingModel
(:212-453).  Modules are parameter containers with the reference's names (so `clip.*` / `hash.*` checkpoints load
strict, including the shared `hash.gcl_i` / `hash.gcl_t` module and the `position.pe` buffers); every forward is native:

  ResidualMLPs                 LayerNorm + GEMM(GELU epilogue) + GEMM(residual epilogue)
  GlobalConceptLearning        ... + Linear(512,K,no bias)+tanh in one launch
  LocalizedTokenAggregation    cmh_mith_lta (mask, per-token top-k, softmax over tokens, weighted token sum)
  PositionalEncoding           cmh_add_positional
  concept Transformer          cmh_transformer_blocks (the same block kernels as the CLIP towers)
  BitwiseHashing               cmh_bitwise_hash
  F.normalize                  cmh_l2_normalize_rows
Layouts: tensors are produced batch-major and returned as views in the reference's layouts ([L,N,D] / [K,N,D]).
The trunk's attention maps are returned as None (MITH.forward discards them, model/MITH.py:464-465)."""
import math

import torch
from torch import nn

import cmh_native as N
import mith_ops as M
from model.base.model import CLIP, Transformer, VisionTransformer, _fill_blocks, convert_weights, no_backward  # noqa: F401
from streams import overlapped


class ViT(VisionTransformer):
    """model/MITH.py:49-82: returns (seq_tokens [g*g, B, E], attn_weight=None, cls_token [B, E])."""


class CLIP1(CLIP):
    def encode_image(self, image):
        from model.base import train_ops as T
        params = T.vit_params(self.visual)
        if not self.assume_frozen and T.wants_grad(params):                 # training: tape-keeping forward, token gradients back
            image = N.f32c(image)
            s = self._vit_struct()
            tok = T.VitTrainTokens.apply(self, image, *params).view(image.shape[0], -1, s.embed_dim)
        else:
            tok = M.vit_encode_tokens(self, image)                       # [B, T, E]
        return tok[:, 1:].permute(1, 0, 2), None, tok[:, 0]

    def encode_text(self, text, key_padding_mask):
        from model.base import train_ops as T
        params = T.text_params(self)
        if not self.assume_frozen and T.wants_grad(params):
            text = text.to(torch.int64).contiguous()
            kpm = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
            tok, rows = T.TextTrainTokens.apply(self, text, kpm, *params)
            tok = tok.view(text.shape[0], text.shape[1], -1)
        else:
            tok, rows = M.text_encode_tokens(self, text, key_padding_mask,   # [B, L, E]
                                             padded_unused=bool(getattr(self, "padded_tokens_unused", False)))
        B, L, E = tok.shape
        new_kpm = key_padding_mask + (text == 49407)                 # model/MITH.py:134 (bool OR)
        eos = tok.reshape(B * L, E)[rows.long()]
        return tok.permute(1, 0, 2), None, new_kpm, eos


def build_model(state_dict: dict):
    """model/MITH.py:163-204 (same shape inference as model/base/model.py::build_model, CLIP1 instead of CLIP)."""
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    embed_dim = state_dict["text_projection"].shape[1]
    tw = state_dict["ln_final.weight"].shape[0]
    layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP1(embed_dim, vision_patch_size * grid_size, vision_layers, vision_width, vision_patch_size,
                  state_dict["positional_embedding"].shape[0], state_dict["token_embedding.weight"].shape[0], tw, tw // 64,
                  layers)
    for key in ["input_resolution", "context_length", "vocab_size"]:
        if key in state_dict:
            del state_dict[key]
    convert_weights(model)
    model.load_state_dict(state_dict)
    return model


def load_download_clip(clip_path) -> nn.Module:
    if isinstance(clip_path, dict):
        return build_model(dict(clip_path))
    try:
        state_dict = torch.jit.load(clip_path, map_location="cpu").eval().state_dict()
    except RuntimeError:
        state_dict = torch.load(clip_path, map_location="cpu")
    return build_model(state_dict)


class ResidualMLPs(nn.Module):
    def __init__(self, org_dim, dropout=0., num_layers=2, activation='relu'):
        super().__init__()
        assert dropout == 0, "MITH runs its residual MLPs with dropout 0 (train/MITH/get_args.py:11)"
        self.num_layers = num_layers
        self.activation = activation
        self.activation_layer = nn.ReLU() if activation == 'relu' else nn.GELU()
        self.mlps = nn.ModuleList(nn.Sequential(nn.Linear(org_dim, 4 * org_dim), self.activation_layer,
                                                nn.Dropout(p=dropout), nn.Linear(4 * org_dim, org_dim))
                                  for _ in range(num_layers))
        self.lns = nn.ModuleList(nn.LayerNorm(org_dim) for _ in range(num_layers))

    def forward(self, x):
        shape = x.shape
        x = N.f32c(x).reshape(-1, shape[-1])
        dt = getattr(self, "_gemm_dt", N.F32)
        lp = dt == N.BF16      # bf16 GEMMs: LayerNorm and the first GEMM hand their result on as the bf16 operand the next GEMM reads
        for i in range(self.num_layers):        # (round 5: two cast passes per layer over [tokens, 512] / [tokens, 2048] gone; same bits)
            h = N.layernorm(x, self.lns[i].weight, self.lns[i].bias, out_bf16=lp)
            u = M.gemm(h, self.mlps[i][0].weight, self.mlps[i][0].bias, act=self.activation, dtype=dt, out_bf16=lp)
            x = M.gemm(u, self.mlps[i][3].weight, self.mlps[i][3].bias, residual=x, dtype=dt)
        return x.reshape(shape)

    def forward_train(self, x):
        """The same layers through autograd Functions (mith_train_ops.py); x [R, D]."""
        import mith_train_ops as T
        from backward_ops import LayerNormFn
        if self.activation != 'gelu':
            raise NotImplementedError("MITH training: only the reference's default activation 'gelu' has a backward")
        dt = getattr(self, "_gemm_dt", N.F32)
        for i in range(self.num_layers):
            h = LayerNormFn.apply(x, self.lns[i].weight, self.lns[i].bias)
            u = T.GeluFn.apply(T.GemmLinear.apply(h, self.mlps[i][0].weight, self.mlps[i][0].bias, dt))
            x = x + T.GemmLinear.apply(u, self.mlps[i][3].weight, self.mlps[i][3].bias, dt)
        return x


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0., max_len=128):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        pe = pe.unsqueeze(0).transpose(0, 1) / (d_model ** 0.5)      # [max_len, 1, d_model]
        self.register_buffer('pe', pe)

    def forward(self, x):
        """x batch-major [N, K, D] (reference: [K, N, D])."""
        return M.add_positional(x, self.pe[:x.shape[1], 0, :])


class BitwiseHashing(nn.Module):
    def __init__(self, org_dim, k_bits=32):
        super().__init__()
        self.k = k_bits
        self.fc_list = nn.ModuleList(nn.Linear(org_dim, 1) for _ in range(k_bits))

    def forward(self, x):
        """x [N, K, D] -> tanh codes [N, K]."""
        w = torch.cat([l.weight for l in self.fc_list], 0)
        b = torch.cat([l.bias for l in self.fc_list], 0)
        return M.bitwise_hash(x, w, b)


class GlobalConceptLearning(nn.Module):
    def __init__(self, k_concept, org_dim, dropout=0., activation='relu', res_mlp_layers=0):
        super().__init__()
        self.mlp = ResidualMLPs(org_dim=org_dim, dropout=dropout, num_layers=res_mlp_layers,
                                activation=activation) if res_mlp_layers != 0 else nn.Identity()
        self.common_concept_embedding = nn.Linear(org_dim, k_concept, bias=False)

    def forward(self, x):
        x = self.mlp(x)
        flat = N.f32c(x).reshape(-1, x.shape[-1])
        c = N.linear_act(flat, self.common_concept_embedding.weight, None, N.ACT_TANH)
        return x, c.reshape(*x.shape[:-1], -1)

    def forward_train(self, x):
        """x [R, D] with gradients: (residual-MLP features, tanh concept logits)."""
        from backward_ops import LinearAct
        x = self.mlp.forward_train(N.f32c(x)) if isinstance(self.mlp, ResidualMLPs) else x
        w = self.common_concept_embedding.weight
        c = LinearAct.apply(x, w, torch.zeros(w.shape[0], device=w.device), N.ACT_TANH, None, 0.0)
        return x, c


class LocalizedTokenAggregation(nn.Module):
    def __init__(self, top_k):
        super().__init__()
        self.top_k = top_k


class LocalConceptTransforming(nn.Module):
    def __init__(self, clip_embed_dim, k_bits, transformer_layers, dropout, top_k):
        super().__init__()
        self.lta = LocalizedTokenAggregation(top_k=top_k)
        self.position = PositionalEncoding(clip_embed_dim, dropout=dropout, max_len=k_bits)
        self.transformer = Transformer(width=clip_embed_dim, layers=transformer_layers, heads=clip_embed_dim // 64)
        self.hashing = BitwiseHashing(org_dim=clip_embed_dim, k_bits=k_bits)
        self._blocks = None

    def forward(self, tokens_bm, sim_bm, l0, L, key_padding_mask=None):
        """tokens_bm [N, Ltot, D], sim_bm [N, Ltot, K] batch-major -> (hash [N,K], transformed concept tokens [N,K,D])."""
        x = M.lta(tokens_bm, sim_bm, key_padding_mask, l0, L, self.lta.top_k)          # [N, K, D]
        x = self.position(x)
        Nb, K, D = x.shape
        keep = []
        dt = getattr(self, "_gemm_dt", N.F32)
        arr = _fill_blocks(self.transformer.resblocks, dt, keep)
        y = M.transformer_blocks(arr, len(self.transformer.resblocks), x.reshape(Nb * K, D), Nb, K, dt).reshape(Nb, K, D)
        del keep
        return self.hashing(y), y

    def forward_train(self, tokens_bm, sim_bm, l0, L, key_padding_mask=None):
        """Same outputs with gradients w.r.t. the tokens and every parameter (the similarities stay detached, as upstream)."""
        import mith_train_ops as T
        x = T.LtaFn.apply(tokens_bm, sim_bm.detach(), key_padding_mask, l0, L, self.lta.top_k)
        x = T.AddPosFn.apply(x, self.position.pe[:x.shape[1], 0, :])
        Nb, K, D = x.shape
        keep = []
        dt = getattr(self, "_gemm_dt", N.F32)
        arr = _fill_blocks(self.transformer.resblocks, dt, keep)
        params = T.blocks_params(self.transformer.resblocks)
        y = T.BlocksTrain.apply(x.reshape(Nb * K, D), Nb, K, dt, arr, keep, *params).reshape(Nb, K, D)
        w = torch.cat([l.weight for l in self.hashing.fc_list], 0)
        b = torch.cat([l.bias for l in self.hashing.fc_list], 0)
        return T.BitHashFn.apply(y, w, b), y


class HashingModel(nn.Module):
    def __init__(self, clip_embed_dim=512, args=None):
        super().__init__()
        self.k_bits = k_bits = args.output_dim
        self.gcl_i = self.gcl_t = GlobalConceptLearning(k_concept=k_bits, org_dim=clip_embed_dim, dropout=args.dropout,
                                                        activation=args.activation, res_mlp_layers=args.res_mlp_layers)
        self.lct_i = LocalConceptTransforming(clip_embed_dim, k_bits, args.transformer_layers, 0, args.top_k_label)
        self.lct_t = LocalConceptTransforming(clip_embed_dim, k_bits, args.transformer_layers, 0, args.top_k_label)
        self.img_concept_proj = nn.Linear(clip_embed_dim, clip_embed_dim)
        self.txt_concept_proj = nn.Linear(clip_embed_dim, clip_embed_dim)

    def set_gemm_dtype(self, name: str):
        """"f32": exact-fp32 MFMA GEMMs (parity mode); "bf16": bf16 operands / f32 accumulate for the ResidualMLPs, the concept
        transformer and the concept projections - the GEMMs are 3/4 of the f32 HashingModel's time (tools/mith_bench.py)."""
        dt = {"f32": N.F32, "fp32": N.F32, "bf16": N.BF16, "bfloat16": N.BF16}[name.lower()]
        for m in self.modules():
            m._gemm_dt = dt
        return self

    def forward(self, img_tokens, txt_tokens, img_cls, txt_eos, key_padding_mask):
        """Reference layouts in (img_tokens [49,N,D], txt_tokens [L,N,D]) and out (trans_tokens_* [K,N,D])."""
        D = self.img_concept_proj.in_features
        for name, t in (("img_tokens", img_tokens), ("txt_tokens", txt_tokens), ("img_cls", img_cls), ("txt_eos", txt_eos)):
            if t.shape[-1] != D:
                raise ValueError(f"HashingModel was built for {D}-d CLIP features, {name} has {t.shape[-1]}")
        if torch.is_grad_enabled() and (img_tokens.requires_grad or self.img_concept_proj.weight.requires_grad):
            return self._forward_train(img_tokens, txt_tokens, img_cls, txt_eos, key_padding_mask)
        dt = getattr(self, "_gemm_dt", N.F32)

        def branch(gcl, lct, proj, cls, tokens, kpm):
            """one modality: the cls-level concepts, the token-level concepts, their aggregation and the concept transformer"""
            res_cls, cls_hash = gcl(cls)
            tok = tokens.permute(1, 0, 2).contiguous()                                   # [N, L, D]
            hash_tok, trans = lct(tok, gcl(tok)[1], 0, tok.shape[1], kpm)
            Nb, K, Dd = trans.shape
            p = M.gemm(trans.reshape(Nb * K, Dd), proj.weight, proj.bias, dtype=dt)
            return cls_hash, M.l2_normalize_rows(res_cls), hash_tok, M.l2_normalize_rows(p).reshape(Nb, K, Dd).permute(1, 0, 2)

        # the image side and the text side meet only in the losses (model/MITH.py:427-453 runs them one after the other): one HIP
        # stream each (round 5; every scratch buffer is per stream, cmh_native.workspace).  gcl_i IS gcl_t (one shared module,
        # model/MITH.py:407): its cached bf16 weight copies are (re)made here, on the caller's stream, before the branches fork
        self._prime_shared_weights(dt)
        bi, bt = overlapped(lambda: branch(self.gcl_i, self.lct_i, self.img_concept_proj, img_cls, img_tokens, None),
                            lambda: branch(self.gcl_t, self.lct_t, self.txt_concept_proj, txt_eos, txt_tokens, key_padding_mask))
        out = {}
        out['img_cls_hash'], out['res_img_cls'], out['img_tokens_hash'], out['trans_tokens_i'] = bi
        out['txt_cls_hash'], out['res_txt_cls'], out['txt_tokens_hash'], out['trans_tokens_t'] = bt
        return out

    def _prime_shared_weights(self, dt):
        """gcl_i IS gcl_t (one shared module, model/MITH.py:407): its cached bf16 weight copies are (re)made on the caller's stream,
        before the two modality branches fork onto their own streams"""
        if dt == N.BF16:
            for m in self.gcl_i.modules():
                if isinstance(m, nn.Linear):
                    M.weight_bf16(m.weight)

    def _forward_train(self, img_tokens, txt_tokens, img_cls, txt_eos, key_padding_mask):
        """forward() through autograd Functions; layouts as in forward().  Like forward(): one HIP stream per modality (autograd
        replays every backward on the stream its forward ran on, so the backward pass of the two branches overlaps the same way)."""
        import mith_train_ops as T
        dt = getattr(self, "_gemm_dt", N.F32)

        def branch(gcl, lct, proj, cls, tokens, kpm):
            res_cls, cls_hash = gcl.forward_train(cls)
            res = T.L2NormFn.apply(res_cls)
            tok = tokens.permute(1, 0, 2).contiguous()
            with torch.no_grad():                                    # gcl(tokens)[1].detach() upstream (:441-442)
                sim = gcl(tok)[1]
            hash_tok, trans = lct.forward_train(tok, sim, 0, tok.shape[1], kpm)
            Nb, K, D = trans.shape
            p = T.GemmLinear.apply(trans.reshape(Nb * K, D), proj.weight, proj.bias, dt)
            return cls_hash, res, hash_tok, T.L2NormFn.apply(p).reshape(Nb, K, D).permute(1, 0, 2)

        self._prime_shared_weights(dt)
        bi, bt = overlapped(lambda: branch(self.gcl_i, self.lct_i, self.img_concept_proj, img_cls, img_tokens, None),
                            lambda: branch(self.gcl_t, self.lct_t, self.txt_concept_proj, txt_eos, txt_tokens, key_padding_mask))
        out = {}
        out['img_cls_hash'], out['res_img_cls'], out['img_tokens_hash'], out['trans_tokens_i'] = bi
        out['txt_cls_hash'], out['res_txt_cls'], out['txt_tokens_hash'], out['trans_tokens_t'] = bt
        return out


class MITH(nn.Module):
    def __init__(self, args=None):
        super(MITH, self).__init__()
        self.args = args
        self.clip = load_download_clip(self.args.clip_path)
        self.hash = HashingModel(clip_embed_dim=512, args=args)
        # HashingModel is the only reader of the text tokens and LocalizedTokenAggregation gives padded positions weight 0
        # (reference model/MITH.py:349-376): the trunk does not compute them (cmh_text_encode_tokens_packed; zeros come back)
        self.clip.padded_tokens_unused = True

    def forward(self, image, text, key_padding_mask):
        (img_tokens, _, img_cls), (txt_tokens, _, new_key_padding_mask, txt_eos) = overlapped(
            lambda: self.clip.encode_image(image), lambda: self.clip.encode_text(text, key_padding_mask))
        return self.hash(img_tokens, txt_tokens, img_cls, txt_eos, new_key_padding_mask)
