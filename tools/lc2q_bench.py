"""The 12-wave GEMM form on e4m3 operands (cmh_set_gemm_lc(7)) against the wide kernel's fp8 launches: bit equality and the kernels' own
durations.  python tools/lc2q_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
DEV = "cuda:0"
g = torch.Generator().manual_seed(2)
def e4m3(shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
N.set_gemm_rows(0)
print(f"{'case':22s} {'wide us':>9s} {'lc2q us':>9s} {'ratio':>7s}  bits")
for name, (M, Nn, K) in {"v_qkv": (12800, 2304, 768), "t_qkv": (10499, 1536, 512), "v_qkv_b512": (25600, 2304, 768), "small": (2049, 256, 512)}.items():
    x, w = e4m3((M, K)), e4m3((Nn, K), K ** -0.5 * 8)
    cs = (torch.rand(Nn, generator=g) * 0.1 + 0.05).to(DEV); b = torch.randn(Nn, generator=g).to(DEV)
    def run(): return N.linear_gemm_fp8(x, w, cs, 0.37, bias=b, out="bf16")
    N.set_gemm_lc(0); ref = run()
    N.set_gemm_lc(7); got = run()
    ok = torch.equal(ref, got)
    t = {}
    for mode in (0, 7, 0, 7):
        N.set_gemm_lc(mode)
        for _ in range(5): run()
        N.prof_gemm_begin(128)
        for _ in range(40): run()
        ms, fl, n = N.prof_gemm_end()
        t.setdefault(mode, []).append(ms * 1e3 / 40)
    # the same launch without its output stores (epi | 256, timing only): what the epilogue's stores cost
    o = torch.empty(M, Nn, dtype=torch.bfloat16, device=DEV)
    N.set_gemm_lc(7)
    def raw(): N.check(N.lib().cmh_linear_gemm_fp8(N.ptr(x), N.ptr(w), N.ptr(cs), 0.37, N.ptr(b), None, N.ptr(o), 1.0, M, Nn, K, 1 | 8 | 256, N.stream_ptr(torch.device(DEV))), "fp8")
    for _ in range(5): raw()
    N.prof_gemm_begin(128)
    for _ in range(40): raw()
    ns_us = N.prof_gemm_end()[0] * 1e3 / 40
    N.set_gemm_lc(0)
    w_us, l_us = min(t[0]), min(t[7])
    print(f"{name:22s} {w_us:9.2f} {l_us:9.2f} {l_us / w_us:7.3f}  {'equal' if ok else 'MISMATCH ' + str(int((ref != got).sum()))}   lc2q without stores {ns_us:7.2f} us", flush=True)
