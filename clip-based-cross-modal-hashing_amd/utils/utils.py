"""The three similarity helpers the DCHMT loss uses (reference utils/utils.py:26-69), on libcmh.

calc_neighbor runs on packed label bits; euclidean/cosine similarity matrices are produced by
torch only as API-parity conveniences (B x B, outside the timed path): the loss itself is fused in
cmh_dchmt_loss and never materialises them."""
import torch

import cmh_native as N

__all__ = ["calc_neighbor", "euclidean_similarity", "cosine_similarity"]


def calc_neighbor(a: torch.Tensor, b: torch.Tensor):
    N.require_gpu(a, b)
    return N.calc_neighbor(N.pack_labels(a), N.pack_labels(b), a.shape[1])


def euclidean_similarity(a: torch.Tensor, b: torch.Tensor):
    return torch.cdist(a, b, p=2.0)


def cosine_similarity(a: torch.Tensor, b: torch.Tensor):
    a = a / a.norm(dim=-1, keepdim=True) if len(torch.where(a != 0)[0]) > 0 else a
    b = b / b.norm(dim=-1, keepdim=True) if len(torch.where(b != 0)[0]) > 0 else b
    return torch.matmul(a, b.t())
