"""Hammer the 12-wave GEMM form (cmh_set_gemm_lc(4)) against the wide kernel: every block shape of both towers, plain and grouped,
many launches each with fresh operands, every output compared with torch.equal.  python tools/lc2_stress.py [--rounds 40]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=40); a = ap.parse_args()
DEV = "cuda:0"
g = torch.Generator().manual_seed(3)
SHAPES = [((12800, 2304, 768), (10499, 1536, 512), 0), ((12800, 768, 768), (10499, 512, 512), 1),
          ((12800, 3072, 768), (10499, 2048, 512), 2), ((12800, 768, 3072), (10499, 512, 2048), 1),
          ((300, 256, 1088), (2049, 512, 1024), 1), ((5000, 1024, 576), (130, 512, 1024), 0)]
def make(M, Nn, K, kind):
    p = {"x": torch.randn(M, K, generator=g).bfloat16().to(DEV), "w": (torch.randn(Nn, K, generator=g) * K ** -0.5).bfloat16().to(DEV),
         "bias": torch.randn(Nn, generator=g).to(DEV)}
    if kind == 1: p["residual"] = torch.randn(M, Nn, generator=g).half().to(DEV)
    return p
def plain(p, kind): return N.linear_gemm(p["x"], p["w"], bias=p["bias"], residual=p.get("residual"), quickgelu=kind == 2, out_bf16=kind != 1, out_f16=kind == 1)
N.set_gemm_rows(0)
bad = total = 0
for rnd in range(a.rounds):
    for sa, sb, kind in SHAPES:
        ps = [make(*sa, kind), make(*sb, kind)]
        N.set_gemm_lc(0)
        ref = [plain(p, kind) for p in ps]
        refg = N.linear_gemm_grouped(ps, quickgelu=kind == 2, out="f16" if kind == 1 else "bf16")
        N.set_gemm_lc(4)
        for rep in range(3):
            got = [plain(p, kind) for p in ps]
            gotg = N.linear_gemm_grouped(ps, quickgelu=kind == 2, out="f16" if kind == 1 else "bf16")
            for r, o in zip(ref + list(refg), got + list(gotg)):
                total += 1
                if not torch.equal(r, o):
                    bad += 1
                    print("MISMATCH", sa, sb, kind, "round", rnd, "rep", rep, int((r != o).sum()), flush=True)
    if rnd % 10 == 9: print(f"round {rnd + 1}: {total} outputs compared, {bad} mismatches", flush=True)
N.set_gemm_lc(0)
print(f"lc2 stress: {total} outputs compared, {bad} mismatches")
sys.exit(1 if bad else 0)
