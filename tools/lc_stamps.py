"""Where the loader / consumer GEMM's MFMA waves spend their time, and at which clock: a -DLC_STAMPS build of csrc/gemm_lc.hip
(make -C clip-based-cross-modal-hashing_amd/csrc BUILD=build_st EXTRA=-DLC_STAMPS; CMH_LIB=.../build_st/libcmh.so).  Per workgroup, MFMA
wave 0: shader-clock ticks of the kernel / until the first stage / inside the K loops / inside the epilogues, and the 100 MHz wall
clock over the same span: clock = ticks / wall."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import numpy as np, torch, cmh_native as N
dev = torch.device("cuda:0")
shapes = {"v_qkv": (12800, 2304, 768, 9), "t_qkv": (10499, 1536, 512, 9), "v_fc1": (12800, 3072, 768, 11), "v_fc2": (12800, 768, 3072, 9),
          "sq4096": (4096, 4096, 4096, 9),
          # few workgroups busy (36 / 72 tiles of the same N, K): is a tile's store tail the CU's own limit or the chip's, all CUs storing at once?
          # epi | 256: the timing-only instantiation that skips the output stores - how much of (dispatch - in-kernel) is dirty lines leaving L2?
          "v_qkv_nostore": (12800, 2304, 768, 9 | 256), "v_fc1_nostore": (12800, 3072, 768, 11 | 256),
          "qkv_36wg": (512, 2304, 768, 9), "qkv_72wg": (1024, 2304, 768, 9), "qkv_144wg": (2048, 2304, 768, 9)}
N.set_gemm_lc(1)
for name, (M, Nn, K, epi) in shapes.items():
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(Nn, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(Nn, device=dev); out = torch.empty(M, Nn, dtype=torch.bfloat16, device=dev)
    for data in ("random", "zeros"):
        if data == "zeros":
            x.zero_(); w.zero_()
        for _ in range(200):        # ~10 ms of back-to-back launches: the clock settles
            N.check(N.lib().cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), N.ptr(b), None, N.ptr(out), M, Nn, K, epi, N.stream_ptr(dev)), "gemm")
        torch.cuda.synchronize()
        N.prof_gemm_begin(64)          # the dispatch's own begin-to-end time (start / stop events of the launch), same launches
        for _ in range(40):
            N.check(N.lib().cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), N.ptr(b), None, N.ptr(out), M, Nn, K, epi, N.stream_ptr(dev)), "gemm")
        disp_us = N.prof_gemm_end()[0] * 1e3 / 40
        buf = np.zeros(256 * 8, dtype=np.uint64)
        assert N.lib().cmh_debug_lc_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        s = buf.reshape(256, 8).astype(np.float64)[:min(256, -(-M // 128) * (Nn // 256))]      # (workgroups this launch did not have keep older stamps)
        s = s[s[:, 6] == s[:, 6].max()]          # the workgroups with the most tiles set the launch's length
        tot, wall, first, kl, ep, ks, tiles, nt = s.mean(0)      # ks: K-steps, nt: tiles of the stamped workgroup
        ghz = tot / (wall * 10.0) if wall else 0.0      # ticks per 10 ns
        print(f"{name:7s} {data:6s}: {tiles:.0f} tiles ({nt:.0f} multiplied by the stamped group, {ks:.0f} K-steps)  kernel {tot:8.0f} ticks = {wall / 100:6.2f} us -> {ghz:4.2f} GHz (dispatch begin-to-end {disp_us:6.2f} us) "
              f"| first stage {first:6.0f} | its K loops {kl:8.0f} ({kl / max(ks, 1):6.0f} per K-step, {kl / max(ks, 1) / ghz / 1e3:5.3f} us) "
              f"| behind them (epilogue set-up / direct epilogue) {ep:7.0f} ({ep / max(nt, 1):6.0f} per tile)", flush=True)
