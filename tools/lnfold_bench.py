"""Per-shape cost of the LayerNorm fold: the producer (residual GEMM + row sums) against the plain residual GEMM, the consumer
(raw fp16 rows, fp16 W * gamma, epilogue normalisation) against the plain bf16 GEMM; back-to-back launches, us per launch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clip-based-cross-modal-hashing_amd"))
import torch
import cmh_native as N

DEV = torch.device("cuda:0")
REP = int(os.environ.get("REP", 40))


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    shapes = [("v", 12800, 768), ("t", 10499, 512)]
    for name, M, d in shapes:
        a = torch.randn(M, d, generator=g).bfloat16().to(DEV)
        a4 = torch.randn(M, 4 * d, generator=g).bfloat16().to(DEV)
        wo = (torch.randn(d, d, generator=g) * d ** -0.5).bfloat16().to(DEV)
        w2 = (torch.randn(d, 4 * d, generator=g) * (4 * d) ** -0.5).bfloat16().to(DEV)
        bo = torch.randn(d, generator=g).to(DEV)
        r = (2.0 * torch.randn(M, d, generator=g)).half().to(DEV)
        x16, part = N.linear_gemm_ln_producer(a, wo, bo, r)
        gamma = (1.0 + 0.3 * torch.randn(d, generator=g)).to(DEV)
        beta = (0.2 * torch.randn(d, generator=g)).to(DEV)
        h = N.layernorm(x16.float(), gamma, beta, out_bf16=True)
        for tag, A, W in (("out  (K=d)", a, wo), ("fc2 (K=4d)", a4, w2)):
            t0 = timeit(lambda: N.linear_gemm(A, W, bias=bo, residual=r, out_f16=True))
            o_, p_ = N.linear_gemm_ln_producer(A, W, bo, r)
            t1 = timeit(lambda: N.linear_gemm_ln_producer(A, W, bo, r, out=o_, part=p_))
            print(f"{name}_{tag:12s} M={M} N={d} K={A.shape[1]}: plain {t0:7.2f} us   +row sums {t1:7.2f} us")
        for tag, No, act in (("qkv", 3 * d, False), ("fc1", 4 * d, True)):
            w = (torch.randn(No, d, generator=g) * d ** -0.5).to(DEV)
            b = torch.randn(No, generator=g).to(DEV)
            wb = w.bfloat16()
            wf, bf, cf = N.ln_fold_weight(w, gamma, beta, b)
            t0 = timeit(lambda: N.linear_gemm(h, wb, bias=b, quickgelu=act, out_bf16=True))
            t1 = timeit(lambda: N.linear_gemm_ln_consumer(x16, part, wf, bf, cf, quickgelu=act))
            print(f"{name}_{tag:12s} M={M} N={No} K={d}: plain {t0:7.2f} us   folded {t1:7.2f} us")


if __name__ == "__main__":
    main()
