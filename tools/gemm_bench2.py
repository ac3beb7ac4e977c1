#!/usr/bin/env python3
"""GEMM micro-benchmark with the encoder's REAL epilogues and the packed text row count, through the C ABI
(cmh_linear_gemm).  Two modes per shape set:
  iso    each GEMM alone, back to back on the same buffers (hot caches)
  chain  the four GEMMs of a block in the encoder's order, repeated (each launch meets the caches the others left)
   python tools/gemm_bench2.py [--iters 50] [--sets vision,text,textdense] [--dtype bf16|fp8]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch  # noqa: E402

import cmh_native as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--sets", default="vision,text,textdense")
ap.add_argument("--text-rows", type=int, default=10499)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"])
ap.add_argument("--configs", default="-1:-1", help="comma list of tile_rows:order_group pairs for cmh_gemm_tuning (-1 = automatic)")
a = ap.parse_args()
dev = torch.device("cuda:0")
st = N.stream_ptr(dev)
E = dict(bias=1, qgelu=2, res=4, obf=8, rf16=64, of16=128)


def block_shapes(M, d):
    rx = E["bias"] | E["res"] | E["rf16"] | E["of16"]
    return [("qkv", M, 3 * d, d, E["bias"] | E["obf"]), ("out", M, d, d, rx),
            ("fc1", M, 4 * d, d, E["bias"] | E["qgelu"] | E["obf"]), ("fc2", M, d, 4 * d, rx)]


SETS = {"vision": block_shapes(12800, 768), "text": block_shapes(a.text_rows, 512), "textdense": block_shapes(19712, 512)}


def make(M, Nn, K, epi):
    if a.dtype == "fp8":
        x = N.fp8_quantize(torch.randn(M, K, device=dev), 4.0 / 448)
        w, cs = N.fp8_quantize_weight(torch.randn(Nn, K, device=dev) * K ** -0.5)
        b = torch.randn(Nn, device=dev)
        res = torch.randn(M, Nn, device=dev).half() if epi & E["res"] else None
        o8 = bool(epi & E["qgelu"])                      # c_fc feeds c_proj: e4m3 output
        out = torch.empty(M, Nn, dtype=torch.uint8 if o8 else (torch.float16 if epi & E["of16"] else torch.bfloat16), device=dev)
        return x, (w, cs), b, res, out
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(Nn, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(Nn, device=dev)
    res = torch.randn(M, Nn, device=dev).half() if epi & E["res"] else None
    out = torch.empty(M, Nn, dtype=torch.float16 if epi & E["of16"] else torch.bfloat16, device=dev)
    return x, w, b, res, out


def run(M, Nn, K, epi, bufs):
    x, w, b, res, out = bufs
    if a.dtype == "fp8":
        e = (epi & ~E["obf"] | 4096) if out.dtype == torch.uint8 else epi
        N.check(N.lib().cmh_linear_gemm_fp8(N.ptr(x), N.ptr(w[0]), N.ptr(w[1]), 4.0 / 448, N.ptr(b), N.ptr(res), N.ptr(out), 0.05,
                                            M, Nn, K, e, st), "gemm_fp8")
        return
    N.check(N.lib().cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(res), N.ptr(out), M, Nn, K, epi, st), "gemm")


def timeit(fn, iters):
    for _ in range(max(5, iters // 4)):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def warm(seconds=1.5):
    """let the clocks settle under an MFMA load before anything is timed (the first timed config otherwise reads ~10 % low)"""
    import time
    x = torch.randn(8192, 4096, device=dev).bfloat16()
    w = torch.randn(4096, 4096, device=dev).bfloat16()
    o = torch.empty(8192, 4096, dtype=torch.bfloat16, device=dev)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            N.check(N.lib().cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), None, None, N.ptr(o), 8192, 4096, 4096, 8, st), "gemm")
        torch.cuda.synchronize()


import statistics
warm()
cfgs = [tuple(int(v) for v in c.split(":")) for c in a.configs.split(",")]
for sname in a.sets.split(","):
    shapes = SETS[sname]
    bufs = [make(M, Nn, K, epi) for _, M, Nn, K, epi in shapes]
    tot_fl = sum(2.0 * M * Nn * K for _, M, Nn, K, _ in shapes)
    res = {}   # (cfg, name) -> [us per round]

    def chain():
        for (name, M, Nn, K, epi), bf in zip(shapes, bufs):
            run(M, Nn, K, epi, bf)
    for rnd in range(a.rounds):                       # configs interleaved inside every round (cdna guide 5.4 rule 24)
        for cfg in cfgs:
            N.gemm_tuning(*cfg)
            for (name, M, Nn, K, epi), bf in zip(shapes, bufs):
                res.setdefault((cfg, name), []).append(timeit(lambda: run(M, Nn, K, epi, bf), a.iters))
            res.setdefault((cfg, "chain"), []).append(timeit(chain, a.iters))
    for cfg in cfgs:
        label = f"{sname}[{cfg[0]}:{cfg[1]}]"
        iso = 0.0
        for name, M, Nn, K, epi in shapes:
            us = statistics.median(res[(cfg, name)])
            iso += us
            print(f"{label:18s} iso   {name:4s} M={M:6d} N={Nn:5d} K={K:5d} {us:8.2f} us (min {min(res[(cfg, name)]):7.2f}) {2.0 * M * Nn * K / us / 1e6:8.1f} TF/s", flush=True)
        us = statistics.median(res[(cfg, "chain")])
        print(f"{label:18s} iso   block {iso:8.2f} us {tot_fl / iso / 1e6:8.1f} TF/s", flush=True)
        print(f"{label:18s} chain block {us:8.2f} us (min {min(res[(cfg, 'chain')]):7.2f}) {tot_fl / us / 1e6:8.1f} TF/s", flush=True)
N.gemm_tuning(-1, -1)
