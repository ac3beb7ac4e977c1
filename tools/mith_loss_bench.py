"""MITH loss terms at configs[2]'s size (batch 256, 64 bit, 80 classes, 10 000-row bank; cls-level InfoNCE 256 x 256 x 512, token-level
256 groups of 64 x 512): us per forward and per forward + backward, and the difference to torch autograd on the same inputs."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "clip-based-cross-modal-hashing_amd"))
import torch
import mith_train_ops as T
dev = "cuda:0"
torch.manual_seed(0)
B, K, C, Mb, D = 256, 64, 80, 10000, 512
bank = torch.randn(Mb, K, device=dev).tanh(); bl = (torch.rand(Mb, C, device=dev) < 0.1).float()
batch = torch.randn(B, K, device=dev).tanh().requires_grad_(True); lab = (torch.rand(B, C, device=dev) < 0.1).float()


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


def bayes_ref(batch):
    s = 0.5 * (bank @ batch.t()).clamp(-64, 64)
    ls = ((bl @ lab.t()) > 0).float()
    return -(ls * s - torch.log(1 + torch.exp(s))).mean()


def fb(f, x):
    def run():
        x.grad = None
        f().backward()
    return run


loss = T.BayesianLossFn.apply(bank, batch, bl, lab); loss.backward(); g = batch.grad.clone(); batch.grad = None
ref = bayes_ref(batch); ref.backward(); gr = batch.grad.clone(); batch.grad = None
print(f"bayesian loss {float(loss):.7f} (torch {float(ref):.7f}), gradient max |diff| / max |g| = {float((g - gr).abs().max() / gr.abs().max()):.2e}")
print(f"  forward {timeit(lambda: T.BayesianLossFn.apply(bank, batch.detach(), bl, lab)):.1f} us, forward + backward {timeit(fb(lambda: T.BayesianLossFn.apply(bank, batch, bl, lab), batch)):.1f} us")
for name, (Rr, G) in {"cls-level InfoNCE 256 x 256": (256, None), "token-level InfoNCE 256 groups of 64": (16384, 64)}.items():
    a = torch.nn.functional.normalize(torch.randn(Rr, D, device=dev), dim=-1).requires_grad_(True)
    b = torch.nn.functional.normalize(torch.randn(Rr, D, device=dev), dim=-1).requires_grad_(True)
    l = T.InfoNceFn.apply(a, b, G, 0.07); l.backward(); ga = a.grad.clone(); a.grad = None; b.grad = None
    Gs = Rr if G is None else G
    sc = torch.einsum("gid,gjd->gij", a.view(-1, Gs, D), b.view(-1, Gs, D)) / 0.07
    tgt = torch.arange(Gs, device=dev).repeat(Rr // Gs)
    lr = 0.5 * (torch.nn.functional.cross_entropy(sc.reshape(Rr, Gs), tgt) + torch.nn.functional.cross_entropy(sc.transpose(1, 2).reshape(Rr, Gs), tgt))
    lr.backward(); gra = a.grad.clone(); a.grad = None; b.grad = None
    print(f"{name}: {float(l):.7f} (torch {float(lr):.7f}), gradient max |diff| / max |g| = {float((ga - gra).abs().max() / gra.abs().max()):.2e}")
    print(f"  forward {timeit(lambda: T.InfoNceFn.apply(a.detach(), b.detach(), G, 0.07)):.1f} us, forward + backward {timeit(fb(lambda: T.InfoNceFn.apply(a, b, G, 0.07), a)):.1f} us")
