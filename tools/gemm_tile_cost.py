#!/usr/bin/env python3
"""What a tile SWITCH of the wide kernel costs, measured from outside: M = 10 240 rows (64 tiles of 160 rows) x N = 1024 R columns is
exactly R rounds of 256 tiles on 256 CUs, so  t(R, nk) = c0 + R * nk * b + (R - 1) * s  with b = one K-step, s = one tile switch
(epilogue of a tile + whatever of the next tile's start it does not hide) and c0 = launch + first stage + last epilogue.  Least squares
over R in 1..4 and K in 256..3072, per epilogue.   python tools/gemm_tile_cost.py [--dtype bf16]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import numpy as np
import torch
import cmh_native as N

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--rows", type=int, default=10240)
ap.add_argument("--epi", default="", help="only the epilogues whose name contains this")
ap.add_argument("--fp8", action="store_true", help="e4m3 operands (K-steps of 128), e4m3 output for the bias / QuickGELU legs")
a = ap.parse_args()
dev = torch.device("cuda:0")
M = a.rows
for epi_name, kw in (("bias", {}), ("bias + QuickGELU", {"quickgelu": True}), ("bias + f32 residual (f32 out)", {"residual": True}),
                     ("bias + f16 residual (f16 out)", {"residual": "f16"})):
    if a.epi and a.epi not in epi_name:
        continue
    rows, ts = [], []
    for R in (1, 2, 3, 4):
        for K in (256, 512, 768, 1024, 1536, 2304, 3072):
            Nn = 1024 * R
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(Nn, K, device=dev) * K ** -0.5).to(torch.bfloat16)
            b = torch.randn(Nn, device=dev)
            k2 = dict(kw)
            obf = True
            res = k2.pop("residual", False)
            if res == "f16":                          # the bf16 mode's fp16 residual stream (out_proj / c_proj launches)
                k2["residual"] = torch.randn(M, Nn, device=dev).half()
                k2["out_f16"] = True
                obf = False
            elif res:
                k2["residual"] = torch.randn(M, Nn, device=dev)
                obf = False
            f = lambda: N.linear_gemm(x, w, bias=b, out_bf16=obf, **k2)
            if a.fp8:
                if not obf or "residual" in k2:
                    continue
                x8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev)
                w8 = torch.randint(0, 120, (Nn, K), dtype=torch.uint8, device=dev)
                cs = torch.ones(Nn, device=dev)
                f = lambda: N.linear_gemm_fp8(x8, w8, cs, 1e-3, bias=b, out="fp8", out_scale=1.0, **k2)
            for _ in range(15):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(a.iters):
                f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            nk = K // (128 if a.fp8 else 64)
            rows.append([1.0, R * nk, R - 1]); ts.append(us)
            print(f"  {epi_name:30s} R={R} K={K:5d}: {us:8.2f} us  ({2.0 * M * Nn * K / us / 1e6:7.1f} TF/s)", flush=True)
    sol, res, *_ = np.linalg.lstsq(np.array(rows), np.array(ts), rcond=None)
    fit = np.array(rows) @ sol
    print(f"{epi_name}: c0 = {sol[0]:.2f} us, K-step b = {sol[1]:.3f} us, tile switch s = {sol[2]:.2f} us  (rms residual {np.sqrt(np.mean((fit - np.array(ts)) ** 2)):.2f} us)", flush=True)
