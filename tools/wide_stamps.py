"""Where the wide GEMM's K loop spends its cycles: per-wave s_memtime sums of (second half, first half, vmcnt wait,
barrier) from a -DW_STAMPS build of gemm_wide.hip (CMH_LIB=.../libcmh_stamps.so).  Stamps cost ~10 % themselves."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import numpy as np, torch, cmh_native as N
dev = torch.device("cuda:0")
shapes = {"v_out": (12800, 768, 768), "v_qkv": (12800, 2304, 768), "v_fc1": (12800, 3072, 768), "v_fc2": (12800, 768, 3072),
          "v_qkv_1round": (4480, 2304, 768), "t_fc1": (19712, 2048, 512)}
for name, (M, Nn, K) in shapes.items():
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(Nn, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(Nn, device=dev); out = torch.empty(M, Nn, dtype=torch.bfloat16, device=dev)
    for epi in (9, 9 | 256):
        for _ in range(5):
            N.check(N.lib().cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), N.ptr(b), None, N.ptr(out), M, Nn, K, epi, N.stream_ptr(dev)), "gemm")
        torch.cuda.synchronize()
        buf = np.zeros(256 * 8 * 4, dtype=np.uint32)
        assert N.lib().cmh_debug_wide_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        s = buf.reshape(256, 8, 4).astype(np.float64)
        tot = s.sum(-1)
        steps = -(-(M // 160 + (M % 160 > 0)) * (Nn // 256) // 256) * (K // 64)
        for grp, sl in (("A", slice(0, 4)), ("B", slice(4, 8))):
            m = s[:, sl].mean((0, 1)); t = tot[:, sl].mean()
            print(f"{name} epi={epi:3d} waves {grp}: total {t:9.0f} ticks (~{t/steps:6.0f}/K-step)  half2 {100*m[0]/t:5.1f}%  half1 {100*m[1]/t:5.1f}%  vmcnt {100*m[2]/t:5.1f}%  barrier {100*m[3]/t:5.1f}%", flush=True)
