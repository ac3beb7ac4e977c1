"""mAP evaluation at the three dataset scales of the reference (utils/calc_utils.py::calc_map_k_matrix), one direction:
MIRFlickr (2000 x 18015... here the bench.py shape 5000 x 15015, 64 bit), MS-COCO (5000 x 117218, 64 bit, k=None),
NUS-WIDE (2100 x 190834, 128 bit) - reference tie order and stable ties."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
dev = torch.device("cuda:0")
only = sys.argv[1:]                # optional: the scales to run
for name, (Q, Nn, K, C) in {"flickr": (5000, 15015, 64, 24), "coco": (5000, 117218, 64, 80), "nuswide": (2100, 190834, 128, 21)}.items():
    if only and name not in only:
        continue
    g = torch.Generator().manual_seed(1)
    rL = (torch.rand(Nn, C, generator=g) < 0.1).float(); qL = (torch.rand(Q, C, generator=g) < 0.1).float()
    W = torch.randn(C, K, generator=g)
    mk = lambda lab: torch.sign(lab @ W + 0.5 * torch.randn(lab.shape[0], K, generator=g) + 1e-3).to(dev)
    r, q = mk(rL), mk(qL)
    rp, qp, rl, ql = N.pack_codes(r), N.pack_codes(q), N.pack_labels(rL.to(dev)), N.pack_labels(qL.to(dev))
    for tie, tname in ((N.TIE_REFERENCE, "reference"), (N.TIE_STABLE, "stable")):
        N.hamming_map(qp, ql, rp, rl, K, C, tie_order=tie)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mp, _, _ = N.hamming_map(qp, ql, rp, rl, K, C, tie_order=tie)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name:8s} Q={Q} N={Nn} K={K} tie={tname:9s}: {dt * 1e3:9.2f} ms  mAP {float(mp):.6f}", flush=True)
