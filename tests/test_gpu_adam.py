"""Fused BertAdam (cmh_bert_adam_step through model/base/optimization.py::BertAdam) against the oracle, which is pinned to
the reference's BertAdam by tests/golden/adam.npz."""
import numpy as np
import pytest
import torch

import adamutil as au

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["trainer", "plain"])
def test_bertadam_mirror_matches_oracle_and_golden(golden, tag):
    from model.base.optimization import BertAdam
    g = golden("adam.npz")
    kw = au.CONFIGS[tag]
    ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in au.mg.tensors(3)]
    half = len(ps) // 2
    opt = BertAdam([{"params": ps[:half], "lr": au.GROUP0_LR}, {"params": ps[half:]}], **kw)
    for s in range(au.mg.STEPS):
        for p, gr in zip(ps, au.mg.grads(3, s)):
            p.grad = gr.clone().to(DEV)
        opt.step()
        np.testing.assert_allclose(np.array(opt.get_lr()), g[f"{tag}_lr_step{s}"], rtol=1e-15)
    o_p, o_g, o_m, o_v = au.run_oracle(tag)
    for i, p in enumerate(ps):
        st = opt.state[p]
        assert st["step"] == au.mg.STEPS and set(st) == {"step", "next_m", "next_v"}      # the reference's state keys
        got = dict(p=p.detach(), m=st["next_m"], v=st["next_v"], g=p.grad)
        for name, ref in (("p", o_p[i]), ("m", o_m[i]), ("v", o_v[i]), ("g", o_g[i])):
            a = got[name].cpu().numpy().reshape(-1)
            np.testing.assert_allclose(a, ref.reshape(-1), rtol=3e-7, atol=au.atol(ref), err_msg=f"{tag} {name}{i} vs oracle")
            np.testing.assert_allclose(au.cut(a), g[f"{tag}_{name}{i}"], rtol=3e-7, atol=au.atol(ref), err_msg=f"{tag} {name}{i} vs golden")


def test_fused_step_edge_cases():
    import cmh_native as N
    from model.base.optimization import BertAdam
    # unaligned views (storage offset of 1 element), a parameter without gradient, and a second step on the same state
    base = torch.randn(1001, device=DEV)
    p = torch.nn.Parameter(base[1:])            # 4-byte aligned only
    q = torch.nn.Parameter(torch.randn(33, device=DEV))
    r = torch.nn.Parameter(torch.randn(5, device=DEV))       # never gets a gradient
    opt = BertAdam([p, q, r], lr=1e-2, b2=0.98, weight_decay=0.1, max_grad_norm=1.0)
    import oracle.adam_oracle as ao
    ref = {id(p): [p.detach().cpu().numpy().copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)],
           id(q): [q.detach().cpu().numpy().copy(), np.zeros(33, np.float32), np.zeros(33, np.float32)]}
    r0 = r.detach().clone()
    for s in range(2):
        for t in (p, q):
            t.grad = torch.randn_like(t) * (3.0 if t is p else 0.01)
        grads = {id(t): t.grad.cpu().numpy().copy() for t in (p, q)}
        opt.step()
        for t in (p, q):
            pp, _, mm, vv = ao.step(ref[id(t)][0], grads[id(t)], ref[id(t)][1], ref[id(t)][2], s, 1e-2, 0.9, 0.98, 1e-6, 0.1, 1.0)
            ref[id(t)] = [pp, mm, vv]
            np.testing.assert_allclose(t.detach().cpu().numpy(), pp, rtol=3e-7, atol=au.atol(pp))
    assert torch.equal(r, r0) and len(opt.state[r]) == 0
    cpu_p = torch.nn.Parameter(torch.randn(4))                # no CPU path: fails loudly
    cpu_p.grad = torch.randn(4)
    with pytest.raises(N.NativeError):
        BertAdam([cpu_p], lr=1e-3).step()


def test_vitb32_sized_step_runs_in_one_launch_pair():
    """302 tensors / 151 M parameters of ViT-B/32-sized CLIP: one fused step, finite results, step counters advanced."""
    from model.base.model import CLIP
    from model.base.optimization import BertAdam
    import recipe
    torch.manual_seed(302)                             # (unseeded gradients made the scalar logit_scale's update round to nothing once in a while)
    clip = CLIP(**recipe.CLIP_VITB32).to(DEV).float()
    params = [p for p in clip.parameters()]
    opt = BertAdam(params, lr=1e-5, warmup=0.1, schedule="warmup_cosine", b2=0.98, t_total=100, weight_decay=0.2)
    before = [p.detach().clone() for p in params[:20]]
    for step in range(2):                              # step 0 of a warm-up schedule has lr = 0 (optimization.py:26-29)
        for p in params:
            p.grad = torch.randn_like(p) * 1e-2
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step()
        e1.record()
        torch.cuda.synchronize()
        if step == 0:
            assert all(torch.equal(b, p) for b, p in zip(before, params[:20]))
    n = sum(p.numel() for p in params)
    print(f"fused BertAdam step: {len(params)} tensors, {n / 1e6:.1f} M parameters, {e0.elapsed_time(e1):.3f} ms "
          f"({n * 32 / e0.elapsed_time(e1) / 1e9:.2f} TB/s of the 32 B/element it must move)")
    assert all(torch.isfinite(p).all() for p in params[:20])
    assert all(not torch.equal(b, p) for b, p in zip(before, params[:20]) if p.numel() > 1)   # (a scalar's 2e-6-sized update may round away)
    assert all(opt.state[p]["step"] == 2 for p in params)
