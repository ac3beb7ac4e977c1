import os, sys
os.environ["CMH_MAP_STAMPS"] = "1"
# usage: python tools/map_stamps.py [N] [bits] [classes]   (CMH_MAP_MODE=lds1|hybrid|global forces a placement where the size allows it)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
dev = torch.device("cuda:0")
Q, Nn = 5000, int(sys.argv[1]) if len(sys.argv) > 1 else 15015
K, C = int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 24
g = torch.Generator().manual_seed(1234)
rL = (torch.rand(Nn, C, generator=g) < 0.15).float(); qL = (torch.rand(Q, C, generator=g) < 0.15).float()
W = torch.randn(C, K, generator=g)
mk = lambda lab: torch.sign(lab @ W + 0.5 * torch.randn(lab.shape[0], K, generator=g) + 1e-3).to(dev)
r, q = mk(rL), mk(qL)
rp, qp, rl, ql = N.pack_codes(r), N.pack_codes(q), N.pack_labels(rL.to(dev)), N.pack_labels(qL.to(dev))
for _ in range(2):
    N.hamming_map(qp, ql, rp, rl, K, C)
torch.cuda.synchronize()
ws = N.workspace(0, dev, "map")
need = N.lib().cmh_map_workspace_bytes(Q, Nn, K, 0)
off = 0 if need <= 4096 + 64 + 32768 else (need - 32768 - 256 - 64)          # all-LDS placements: the stamps are the workspace; else behind the slices
base = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr() + off
area = ws[base:base + 3072 * 8]
area.zero_()                   # the stamp area: levels the run never reaches must read as zero, not as whatever the buffer held
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); N.hamming_map(qp, ql, rp, rl, K, C); e1.record(); torch.cuda.synchronize()
full = area.view(torch.int64).cpu().numpy()
st = full[:6]
d = [int(st[i + 1] - st[i]) for i in range(5)]
print("one direction: %.3f ms; query 0 of workgroup 0, cycles per phase (shader clock, ~2.1 GHz):" % e0.elapsed_time(e1))
for name, v in zip(["keys", "bfs", "parked-seq", "leaf", "ap"], d):
    print(f"  {name:10s} {v:10d}")
# a level was reached iff its stamp lies inside the breadth-first phase of this query
lv = [(int(full[8 + 2 * i]), int(full[9 + 2 * i])) for i in range(28)
      if st[1] <= full[8 + 2 * i] <= st[2] and 0 < full[9 + 2 * i] < 100000]
for i, (t, n) in enumerate(lv):
    nxt = lv[i + 1][0] if i + 1 < len(lv) else int(st[2])
    print(f"  level {i:2d}: {n:5d} segments {nxt - t:9d} cycles")
for i in range(min(len(lv), 16)):
    b = full[64 + 8 * i: 72 + 8 * i]
    t0 = lv[i][0]
    if b[1]:
        print(f"  level {i:2d}: tasks {int(b[5]):4d} chunks {int(b[6]):3d}  build {int(b[0]-t0):6d}  A {int(b[1]-b[0]):6d}  B {int(b[2]-b[1]):6d}  C {int(b[3]-b[2]):6d}  D {int(b[4]-b[3]):6d}")
    else:
        print(f"  level {i:2d}: tasks {int(b[5]):4d} chunks {int(b[6]):3d}  build {int(b[0]-t0):6d}  A {int(b[4]-b[0]):6d}")
