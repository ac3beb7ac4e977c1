"""Placement A/B of the ranking kernel at one size (run once per CMH_MAP_MODE): python tools/map_mode_ab.py Q N bits"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
Q, Nn, K = (int(v) for v in sys.argv[1:4]); C = 24
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rL = (torch.rand(Nn, C, generator=g) < 0.1).float(); qL = (torch.rand(Q, C, generator=g) < 0.1).float()
W = torch.randn(C, K, generator=g)
mk = lambda lab: torch.sign(lab @ W + 0.5 * torch.randn(lab.shape[0], K, generator=g) + 1e-3).to(dev)
r, q = mk(rL), mk(qL)
rp, qp, rl, ql = N.pack_codes(r), N.pack_codes(q), N.pack_labels(rL.to(dev)), N.pack_labels(qL.to(dev))
for _ in range(2): N.hamming_map(qp, ql, rp, rl, K, C)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): mp, ap, _ = N.hamming_map(qp, ql, rp, rl, K, C)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
import hashlib
print(f"mode {os.environ.get('CMH_MAP_MODE', 'auto'):7s} Q={Q} N={Nn} K={K}: {dt * 1e3:8.3f} ms  mAP {float(mp):.7f}  ap sha {hashlib.sha256(ap.cpu().numpy().tobytes()).hexdigest()[:12]}")
