#!/usr/bin/env python3
"""A/B of the loader / consumer GEMM (csrc/gemm_lc.hip) against the wide kernel, one process, interleaved rounds, the kernels' own
durations (cmh_prof_gemm_*: the dispatch's begin / end).  Shapes: the four GEMMs of a block of both towers at batch 256 (packed
text rows), as plain launches and as the grouped launches of the pair path.
   python tools/lc_bench.py [--iters 30] [--rounds 3] [--check]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch  # noqa: E402

import cmh_native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--check", action="store_true", help="compare the two kernels' outputs (torch.equal)")
ap.add_argument("--only", default="")
ap.add_argument("--mode", type=int, default=1, help="cmh_set_gemm_lc mode of the lc column: 1 the 8-wave kernel, 4 the 12-wave form (takes only the residual-free launches)")
ap.add_argument("--nostore", action="store_true", help="timing-only ablation (epi | 256): skip the output stores of the lc kernel")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)

BLOCK = {   # name: (image (M, N, K), text (M, N, K), kind)   kind 0 bias, 1 fp16 residual stream, 2 QuickGELU
    "qkv": ((12800, 2304, 768), (10499, 1536, 512), 0),
    "out": ((12800, 768, 768), (10499, 512, 512), 1),
    "fc1": ((12800, 3072, 768), (10499, 2048, 512), 2),
    "fc2": ((12800, 768, 3072), (10499, 512, 2048), 1),
}


def make(M, Nn, K, kind):
    p = {"x": torch.randn(M, K, generator=g).bfloat16().to(dev), "w": (torch.randn(Nn, K, generator=g) * K ** -0.5).bfloat16().to(dev),
         "bias": torch.randn(Nn, generator=g).to(dev)}
    if kind == 1:
        p["residual"] = torch.randn(M, Nn, generator=g).half().to(dev)
    return p


def run_plain(p, kind):
    return N.linear_gemm(p["x"], p["w"], bias=p["bias"], residual=p.get("residual"), quickgelu=kind == 2, out_bf16=kind != 1, out_f16=kind == 1)


def run_grouped(ps, kind):
    return N.linear_gemm_grouped(ps, quickgelu=kind == 2, out="f16" if kind == 1 else "bf16")


def timed(fn, iters):
    for _ in range(5):
        fn()
    N.prof_gemm_begin(iters * 2 + 8)
    for _ in range(iters):
        fn()
    ms, fl, n = N.prof_gemm_end()
    return ms * 1e3 / iters, fl / iters, n / iters


cases = []
for name, (im, tx, kind) in BLOCK.items():
    pi, pt = make(*im, kind), make(*tx, kind)
    cases.append((f"v_{name}", lambda pi=pi, kind=kind: run_plain(pi, kind), 2.0 * im[0] * im[1] * im[2]))
    cases.append((f"t_{name}", lambda pt=pt, kind=kind: run_plain(pt, kind), 2.0 * tx[0] * tx[1] * tx[2]))
    cases.append((f"g_{name}", lambda pi=pi, pt=pt, kind=kind: run_grouped([pi, pt], kind),
                  2.0 * im[0] * im[1] * im[2] + 2.0 * tx[0] * tx[1] * tx[2]))

print(f"{'case':8s} {'wide us':>9s} {'lc us':>9s} {'lc/wide':>8s} {'wide TF/s':>10s} {'lc TF/s':>9s}  launches w/lc")
tot = {0: 0.0, 1: 0.0}
for name, fn, flops in cases:
    if a.only and name not in a.only.split(","):
        continue
    if a.check:
        N.set_gemm_lc(0)
        ref = fn()
        N.set_gemm_lc(a.mode)
        got = fn()
        ref = ref if isinstance(ref, (list, tuple)) else [ref]
        got = got if isinstance(got, (list, tuple)) else [got]
        ok = all(torch.equal(r, o) for r, o in zip(ref, got))
        if not ok:
            d = max(float((r.float() - o.float()).abs().max()) for r, o in zip(ref, got))
            bad = sum(int((r != o).sum()) for r, o in zip(ref, got))
            print(f"{name}: MISMATCH max|d| = {d:.4g}, {bad} elements", flush=True)
        else:
            print(f"{name}: bits equal", flush=True)
    best = {0: [], 1: []}
    nl = {}
    for _ in range(a.rounds):
        for mode in (0, 1):
            N.set_gemm_lc(a.mode if mode else 0)
            us, fl, n = timed(fn, a.iters)
            best[mode].append(us)
            nl[mode] = n
    N.set_gemm_lc(0)
    w, l = sorted(best[0])[len(best[0]) // 2], sorted(best[1])[len(best[1]) // 2]
    if name.startswith("g_") or True:
        pass
    print(f"{name:8s} {w:9.2f} {l:9.2f} {l / w:8.3f} {flops / w / 1e6:10.1f} {flops / l / 1e6:9.1f}  {nl[0]:.0f}/{nl[1]:.0f}", flush=True)
