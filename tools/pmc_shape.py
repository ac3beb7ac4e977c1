"""One encoder GEMM shape with its real epilogue, a few launches (for tools/pmc_gemm_shapes.sh: one rocprofv3 --pmc pass per shape and
counter set, so that every counter row belongs to a known shape).   python tools/pmc_shape.py <name> [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "clip-based-cross-modal-hashing_amd"))
import torch, cmh_native as N
dev = torch.device("cuda:0")
E = dict(bias=1, qgelu=2, res=4, obf=8, rf16=64, of16=128)
rx = E["bias"] | E["res"] | E["rf16"] | E["of16"]
T = int(os.environ.get("TEXT_ROWS", "10499"))
SHAPES = {"v_qkv": (12800, 2304, 768, E["bias"] | E["obf"]), "v_out": (12800, 768, 768, rx), "v_fc1": (12800, 3072, 768, E["bias"] | E["qgelu"] | E["obf"]),
          "v_fc2": (12800, 768, 3072, rx), "t_qkv": (T, 1536, 512, E["bias"] | E["obf"]), "t_out": (T, 512, 512, rx),
          "t_fc1": (T, 2048, 512, E["bias"] | E["qgelu"] | E["obf"]), "t_fc2": (T, 512, 2048, rx)}
name, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6
if name in SHAPES:
    M, Nn, K, epi = SHAPES[name]
else:                              # "MxNxK": bf16 output, bias epilogue (tools/pmc_traffic_split.sh)
    M, Nn, K = (int(v) for v in name.split("x"))
    epi = E["bias"] | E["obf"]
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).to(dev).bfloat16()
w = (torch.randn(Nn, K, generator=g) * K ** -0.5).to(dev).bfloat16()
b = torch.randn(Nn, generator=g).to(dev)
res = torch.randn(M, Nn, generator=g).to(dev).half() if epi & E["res"] else None
out = torch.empty(M, Nn, dtype=torch.float16 if epi & E["of16"] else torch.bfloat16, device=dev)
for _ in range(n):
    N.check(N.lib().cmh_linear_gemm(N.BF16, N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(res), N.ptr(out), M, Nn, K, epi, N.stream_ptr(dev)), "gemm")
torch.cuda.synchronize()
print(name, M, Nn, K, epi)
