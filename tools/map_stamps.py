import os, sys
os.environ["CMH_MAP_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
dev = torch.device("cuda:0")
Q, Nn, K, C = 5000, int(sys.argv[1]) if len(sys.argv) > 1 else 15015, 64, 24
g = torch.Generator().manual_seed(1234)
rL = (torch.rand(Nn, C, generator=g) < 0.15).float(); qL = (torch.rand(Q, C, generator=g) < 0.15).float()
W = torch.randn(C, K, generator=g)
mk = lambda lab: torch.sign(lab @ W + 0.5 * torch.randn(lab.shape[0], K, generator=g) + 1e-3).to(dev)
r, q = mk(rL), mk(qL)
rp, qp, rl, ql = N.pack_codes(r), N.pack_codes(q), N.pack_labels(rL.to(dev)), N.pack_labels(qL.to(dev))
for _ in range(2):
    N.hamming_map(qp, ql, rp, rl, K, C)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); N.hamming_map(qp, ql, rp, rl, K, C); e1.record(); torch.cuda.synchronize()
ws = N.workspace(0, dev, "map")
st = ws[:64].view(torch.int64).cpu().numpy()[:6]
d = [int(st[i + 1] - st[i]) for i in range(5)]
print("one direction: %.3f ms; query 0 of workgroup 0, cycles per phase (100 MHz ticks?):" % e0.elapsed_time(e1))
for name, v in zip(["keys", "bfs", "parked-seq", "leaf", "ap"], d):
    print(f"  {name:10s} {v:10d}")
