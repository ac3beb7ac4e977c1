"""Evaluation API of the reference (utils/calc_utils.py) on libcmh.so.

  calc_neighbor      :4-5,:42-45
  calc_hammingDist   :8-13
  calc_map_k_matrix  :16-39   — imported by trainers as `calc_map_k` (train/base.py:11)
Inputs are what the reference passes: f32 codes in {-1,0,+1} and f32 multi-hot labels on any
device (the reference's labels live on the CPU, train/base.py:84-85); they are moved to the code
tensor's GPU, bit-packed, ranked and scored there.  The ranking reproduces the reference's
torch.sort tie order exactly (csrc/hamming_map.hip).  Output: 0-dim f32 tensor like upstream."""
import torch

import cmh_native as N


def _dev(*ts):
    for t in ts:
        if t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise N.NativeError("calc_utils needs a GPU: libcmh has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def calc_neighbor(label1, label2):
    dev = _dev(label1, label2)
    l1, l2 = label1.to(dev).float(), label2.to(dev).float()
    return N.calc_neighbor(N.pack_labels(l1), N.pack_labels(l2), l1.shape[1])


def calc_hammingDist(B1, B2):
    q = B2.shape[1]
    if len(B1.shape) < 2:
        B1 = B1.unsqueeze(0)
    dev = _dev(B1, B2)
    return N.hamming_dist(N.pack_codes(B1.to(dev).float()), N.pack_codes(B2.to(dev).float()), q)


def calc_map_k_matrix(qB, rB, query_L, retrieval_L, k=None, rank=0, return_ap=False, tie_order="reference"):
    """tie_order="reference": the ranking of the reference's torch.sort on the CPU (libstdc++ introsort tie order, bit-exact);
    "stable": ties by ascending database index (~4x faster, mAP moves in the 4th digit: not the reference's number)."""
    dev = _dev(qB, rB)
    qB, rB = qB.to(dev).float(), rB.to(dev).float()
    qL, rL = query_L.to(dev).float(), retrieval_L.to(dev).float()
    bits, classes = rB.shape[1], rL.shape[1]
    mp, ap, _ = N.hamming_map(N.pack_codes(qB), N.pack_labels(qL), N.pack_codes(rB), N.pack_labels(rL),
                              bits, classes, topk=k,
                              tie_order={"reference": N.TIE_REFERENCE, "stable": N.TIE_STABLE}[tie_order])
    mp = mp.cpu()          # the reference returns a CPU scalar (it evaluates on the CPU)
    return (mp, ap) if return_ap else mp
