"""Where a wide-GEMM LAUNCH spends its wall time: per-workgroup s_memrealtime (100 MHz) stamps from a -DW_TIMELINE build
(make -C clip-based-cross-modal-hashing_amd/csrc BUILD=build_tl EXTRA=-DW_TIMELINE; CMH_LIB=.../build_tl/libcmh.so).
Columns are microseconds after the first workgroup's entry: entry / first stage landed / first tile's K loop done / last
tile's K loop done / last epilogue's stores issued / all memory operations complete; min, median, max over workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import numpy as np, torch, cmh_native as N
dev = torch.device("cuda:0")
E = dict(bias=1, qgelu=2, res=4, obf=8, rf16=64, of16=128)
rx = E["bias"] | E["res"] | E["rf16"] | E["of16"]
T = int(os.environ.get("TEXT_ROWS", "10499"))
shapes = {"v_qkv": (12800, 2304, 768, 9), "v_out": (12800, 768, 768, rx), "v_fc1": (12800, 3072, 768, 11), "v_fc2": (12800, 768, 3072, rx),
          "t_qkv": (T, 1536, 512, 9), "t_out": (T, 512, 512, rx), "t_fc1": (T, 2048, 512, 11), "t_fc2": (T, 512, 2048, rx)}
if os.environ.get("VARIANTS"):   # epilogue ablations on one shape: gelu on/off, stores skipped (256) / issued from the epilogue (512)
    shapes = {f"v_fc1 epi={e}": (12800, 3072, 768, e) for e in (11, 9, 11 | 256, 9 | 256, 11 | 512)}
    shapes.update({f"v_qkv epi={e}": (12800, 2304, 768, e) for e in (9, 9 | 256, 9 | 512)})
lib = N.lib()
for name, (M, Nn, K, epi) in shapes.items():
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(Nn, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(Nn, device=dev)
    res = torch.randn(M, Nn, device=dev).half() if epi & 4 else None
    out = torch.empty(M, Nn, dtype=torch.float16 if epi & 128 else torch.bfloat16, device=dev)
    run = lambda: N.check(lib.cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(res), N.ptr(out), M, Nn, K, epi, N.stream_ptr(dev)), "gemm")
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    rows = []
    for rep in range(5):
        lib.cmh_debug_wide_timeline_clear()
        torch.cuda.synchronize()
        run(); torch.cuda.synchronize()
        buf = np.zeros(256 * 8, dtype=np.uint64)
        assert lib.cmh_debug_wide_timeline(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        t = buf.reshape(256, 8)[:, :6].astype(np.float64)
        t = t[(t > 0).all(axis=1)]           # workgroups that ran (a workgroup without a tile returns before its first stamp)
        t = (t - t[:, 0].min()) / 100.0
        rows.append(t)
    t = rows[-1]
    fmt = lambda c: f"{np.min(t[:, c]):6.2f}/{np.median(t[:, c]):6.2f}/{np.max(t[:, c]):6.2f}"
    print(f"{name} M={M} N={Nn} K={K}: {us:6.2f} us/launch back-to-back, {len(t)} workgroups\n   entry {fmt(0)}  stage0 {fmt(1)}  tile0 {fmt(2)}  lastK {fmt(3)}  stores {fmt(4)}  done {fmt(5)}", flush=True)
