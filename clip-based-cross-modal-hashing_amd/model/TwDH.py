"""TwDH model (reference model/TwDH.py:9-171) on libcmh.

ModalityHash keeps the reference's parameters (`atten` = nn.MultiheadAttention over a length-1 sequence, `norm`,
`fc2`) so checkpoints load strict.  With one key the softmax is 1, so attention == out_proj(v_proj(x)): two GEMMs.
The image head's BatchNorm1d runs with BATCH statistics even at eval time, exactly like upstream (Baseclip.eval()
only toggles image_hash/text_hash, never img_hash/txt_hash) — codes therefore depend on batch
composition.  Its running statistics move on every forward, as upstream's do (they are never read, but they are part of the checkpoint).
Centres / transition matrices: `long_center` [C,K] +-1, `short_center` {S: [C,S]}, `trans` {S: [2K,2S]} as tensors or
paths to the reference's .pkl assets (train/TwDH/center/<dataset>/...)."""
import logging
import os

import torch
import torch.nn as nn

import cmh_native as N
from model.modelbase import Baseclip, weights_init_kaiming
from streams import overlapped


def softmax_hash(embed, return_vector=True):
    """Pair softmax over [..., 2] (reference :9-19)."""
    B = embed.shape[0]
    flat = embed.reshape(B, -1)
    out = N.pair_softmax(flat)
    return out if return_vector else out.view(B, -1, 2)


class LayerNorm(nn.LayerNorm):
    pass


class ModalityHash(nn.Module):

    def __init__(self, inputDim=2048, outputDim=64, num_heads=8, batch_first=True, layernorm=True, hash_func=None):
        super(ModalityHash, self).__init__()
        assert hash_func == "softmax", "only the softmax hash function is built (the reference's default)"
        self.bit = outputDim
        self.atten = nn.MultiheadAttention(inputDim, num_heads=num_heads, batch_first=batch_first)
        self.norm = LayerNorm(inputDim) if layernorm else nn.BatchNorm1d(inputDim)
        self.fc2 = nn.Linear(inputDim, outputDim * 2)
        self.fc2.apply(weights_init_kaiming)

    def quantization(self, code):
        return softmax_hash(code)

    def _update_running(self, embed):
        """nn.BatchNorm1d's side effect in training mode (upstream never leaves it): move the running statistics."""
        n = self.norm
        if n.track_running_stats and n.running_mean is not None:
            n.num_batches_tracked += 1
            mom = n.momentum if n.momentum is not None else 1.0 / float(n.num_batches_tracked)
            N.check(N.lib().cmh_batchnorm1d_update_running(N.ptr(embed), float(mom), N.ptr(n.running_mean), N.ptr(n.running_var),
                                                           embed.shape[0], embed.shape[1], N.stream_ptr(embed.device)),
                    "cmh_batchnorm1d_update_running")

    def _forward_train(self, data):
        """The same chain through autograd Functions (each a libcmh forward + backward): v_proj -> out_proj -> norm -> fc2 + ReLU
        -> pair softmax.  The Q / K rows of in_proj receive zero gradients (one key: the softmax is constant), as upstream."""
        from backward_ops import BatchNorm1dTrain, LayerNormFn, LinearAct, PairSoftmax
        d = data.shape[1]
        w_in, b_in = self.atten.in_proj_weight, self.atten.in_proj_bias
        v = LinearAct.apply(data, w_in[2 * d:], b_in[2 * d:], N.ACT_NONE, None, 0.0)
        embed = LinearAct.apply(v, self.atten.out_proj.weight, self.atten.out_proj.bias, N.ACT_NONE, None, 0.0)
        if isinstance(self.norm, nn.BatchNorm1d):
            self._update_running(embed.detach())
            embed = BatchNorm1dTrain.apply(embed, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            embed = LayerNormFn.apply(embed, self.norm.weight, self.norm.bias)
        embed = LinearAct.apply(embed, self.fc2.weight, self.fc2.bias, N.ACT_RELU, None, 0.0)
        return PairSoftmax.apply(embed)

    def forward(self, data):
        if torch.is_grad_enabled() and (data.requires_grad or self.fc2.weight.requires_grad):
            return self._forward_train(data)
        d = data.shape[1]
        w_in, b_in = self.atten.in_proj_weight, self.atten.in_proj_bias
        v = N.linear_act(data, w_in[2 * d:], b_in[2 * d:], N.ACT_NONE)                  # v_proj
        embed = N.linear_act(v, self.atten.out_proj.weight, self.atten.out_proj.bias, N.ACT_NONE)
        if isinstance(self.norm, nn.BatchNorm1d):
            self._update_running(embed)
            embed = N.batchnorm1d_train(embed, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            embed = N.layernorm(embed, self.norm.weight, self.norm.bias)
        embed = N.linear_act(embed, self.fc2.weight, self.fc2.bias, N.ACT_RELU)
        return N.pair_softmax(embed)


def _load_tensor(x):
    return (torch.load(x) if isinstance(x, str) else x).float()


def _load_dict(x):
    if isinstance(x, dict):
        return {str(k): v.float() for k, v in x.items()}
    if os.path.isfile(x):
        return {os.path.basename(x).strip().split(".")[0]: torch.load(x).float()}
    return {item.strip().split(".")[0]: torch.load(os.path.join(x, item)).float() for item in os.listdir(x)}


class MTwDH(Baseclip):

    def __init__(self, outputDim=64, clipPath="./ViT-B-32.pt", writer=None, saveDir="./result/log",
                 logger: logging.Logger = None, is_train=True,
                 long_center="./TwDH/center/coco/long", short_center="./TwDH/center/coco/short",
                 trans="./TwDH/center/coco/trans", num_heads=8, batch_first=True, hash_func: str = "softmax",
                 quan_alpha: float = 0.5, low_rate: float = 0):
        super(MTwDH, self).__init__(outputDim=outputDim, clipPath=clipPath, writer=writer,
                                    saveDir=saveDir, logger=logger, is_train=is_train)
        long_dim = outputDim
        if isinstance(long_center, str):
            long_center = os.path.join(long_center, str(long_dim) + ".pkl")
        if isinstance(trans, str):
            trans = os.path.join(trans, str(long_dim))
        self.img_hash = ModalityHash(inputDim=self.embedDim, outputDim=long_dim, layernorm=False, num_heads=num_heads,
                                     batch_first=batch_first, hash_func=hash_func)
        self.txt_hash = ModalityHash(inputDim=self.embedDim, outputDim=long_dim, layernorm=True, num_heads=num_heads,
                                     batch_first=batch_first, hash_func=hash_func)
        self.long_center = _load_tensor(long_center)
        self.short_center = _load_dict(short_center)
        self.trans = _load_dict(trans)
        self._trans_t = {}
        self.quan_alpha = quan_alpha
        self.low_rate = low_rate
        self.short_dims = [int(k) for k in self.short_center]

    def get_short_dims(self):
        return self.short_dims

    def _short(self, head, long_hash):
        out = {}
        for k, v in self.trans.items():
            key = (k, str(long_hash.device))
            if key not in self._trans_t:                       # [2K,2S] -> Linear layout [2S,2K], once per device
                self._trans_t[key] = v.to(long_hash.device).t().contiguous()
            if torch.is_grad_enabled() and long_hash.requires_grad:
                from backward_ops import LinearAct, PairSoftmax
                w = self._trans_t[key]
                zero_b = torch.zeros(w.shape[0], device=w.device)
                out[k] = PairSoftmax.apply(LinearAct.apply(long_hash, w, zero_b, N.ACT_NONE, None, 0.0))
            else:
                z = N.linear_act(long_hash.detach(), self._trans_t[key], None, N.ACT_NONE)
                out[k] = head.quantization(z)
        return out

    def encode_image(self, image):
        long_hash = self.img_hash(self.clip.encode_image(image))
        return long_hash, self._short(self.img_hash, long_hash)

    def encode_text(self, text):
        long_hash = self.txt_hash(self.clip.encode_text(text))
        return long_hash, self._short(self.txt_hash, long_hash)

    def forward(self, image, text):
        (img_long_hash, img_short_hash), (txt_long_hash, txt_short_hash) = overlapped(
            lambda: self.encode_image(image), lambda: self.encode_text(text))
        return img_long_hash, img_short_hash, txt_long_hash, txt_short_hash, self.long_center, self.short_center
