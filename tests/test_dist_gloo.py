"""world_size-2 gloo test of the multi-GPU exchange step (CPU tensors; the kernels themselves are covered
by the -m gpu tests): fused all-gather of code blocks, ragged shards, query-sharded AP gathered in query
order, index-scatter into the global buffers."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    from conftest import PKG  # noqa: F401  (puts the package on sys.path)
    import dist_utils as du
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = du.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n, K, C = 37, 16, 5                                   # ragged: 19 + 18
    g = torch.Generator().manual_seed(0)
    codes_i = torch.sign(torch.randn(n, K, generator=g))
    codes_t = torch.sign(torch.randn(n, K, generator=g))
    labels = (torch.rand(n, C, generator=g) < 0.3).float()
    index = torch.randperm(n, generator=g)
    lo, hi = du.shard_range(n, rank, world)
    counts = [du.shard_range(n, q, world)[1] - du.shard_range(n, q, world)[0] for q in range(world)]
    fused, widths = du.fuse_columns(codes_i[lo:hi], codes_t[lo:hi], labels[lo:hi], index[lo:hi].float().unsqueeze(1))
    allb = du.all_gather_rows(fused, counts)
    gi, gt, gl, gidx = du.split_columns(allb, widths)
    assert torch.equal(gi, codes_i) and torch.equal(gt, codes_t) and torch.equal(gl, labels)
    buf = torch.zeros(n, K)
    du.scatter_by_index(buf, gidx.squeeze(1), gi)
    ref = torch.zeros(n, K)
    ref[index] = codes_i
    assert torch.equal(buf, ref)
    assert du.row_counts(hi - lo, "cpu") == counts
    # equal shards take the single-collective path
    eq = du.all_gather_rows(codes_i[rank * 10:(rank + 1) * 10])
    assert torch.equal(eq, codes_i[:20])
    # gradient averaging in flat buckets (ragged tensor sizes, several buckets)
    gs = [torch.full((5, 3), float(rank + 1)), torch.arange(7, dtype=torch.float32) * (rank + 1), torch.full((1,), 10.0 * rank)]
    du.allreduce_mean_(gs, bucket_bytes=64)
    assert torch.equal(gs[0], torch.full((5, 3), 1.5)) and torch.equal(gs[1], torch.arange(7, dtype=torch.float32) * 1.5)
    assert torch.equal(gs[2], torch.full((1,), 5.0))
    # the same means from inside backward: three groups, a parameter that never gets a gradient, one that appears late
    class Clip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.visual = torch.nn.Linear(4, 3)
            self.txt = torch.nn.Linear(4, 3)
            self.logit_scale = torch.nn.Parameter(torch.ones([]))          # never used: no gradient, like upstream's

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.clip = Clip()
            self.head = torch.nn.Linear(3, 2)
            self.sometimes = torch.nn.Parameter(torch.ones(2))

    torch.manual_seed(3 + rank)                               # replicas drawn differently, then made equal to rank 0's
    model, twin = Model(), Model()
    model.register_buffer("steps", torch.tensor([rank + 5]))
    du.broadcast_modules_([model])
    mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(both, mine)
    assert torch.equal(both[0], both[1]) and int(model.steps) == 5
    del model.steps
    twin.load_state_dict(model.state_dict())
    sync = du.GradSync.for_model(model)
    assert [len(g) for g in sync.groups] == [3, 2, 3]
    sent_early = []
    for step in range(4):
        x = torch.randn(6, 4, generator=g) * (rank + 1)
        for m in (model, twin):
            m.zero_grad()
            y = m.head(m.clip.visual(x) + m.clip.txt(x)).sum(0)
            if step >= 2:
                y = y * m.sometimes
            y.sum().backward()
            if m is model:
                sent_early.append(sum(sync._sent))
                sync.finish()
            else:
                du.allreduce_mean_([p.grad for p in m.parameters() if p.grad is not None])
        for (name, p), q in zip(model.named_parameters(), twin.parameters()):
            assert (p.grad is None) == (q.grad is None), name
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), (step, name)
    # step 0 learns the pattern; steps 1 and 3 send all three groups from the hooks; in step 2 `sometimes` is new, so its
    # group no longer matches what was learnt and leaves from finish()
    assert sent_early == [0, 3, 2, 3], sent_early
    assert model.clip.logit_scale.grad is None
    sync.remove()
    # query-sharded AP, gathered in query order, summed like the reference
    ap = torch.rand(n, generator=g)
    full = du.gather_query_sharded_ap(ap[lo:hi], n)
    assert torch.equal(full, ap)
    m = du.mean_in_query_order(full)
    acc = np.float32(0)
    for v in ap.numpy():
        acc = np.float32(acc + v)
    assert float(m) == float(acc / np.float32(n))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def _pair_loss(hi, ht, lab, proxies):
    """A stand-in with the structure of the reference's losses (HyP, train/DSPH/loss.py:33-66): a per-sample proxy term plus
    pairwise terms over the whole batch, which is what makes the global batch matter."""
    import torch.nn.functional as F
    cos_p = F.normalize(hi, dim=1) @ F.normalize(proxies, dim=1).t()
    pos = ((1 - cos_p) * lab).sum() / lab.sum().clamp(min=1)
    sim = (lab @ lab.t() > 0).float()
    cross = F.normalize(hi, dim=1) @ F.normalize(ht, dim=1).t()
    neg = (torch.relu(cross) * (1 - sim)).sum() / (1 - sim).sum().clamp(min=1)
    return pos + neg + (torch.cdist(hi, ht) * sim).mean()


def _global_loss_worker(rank, world, port, out_dir):
    from conftest import PKG  # noqa: F401
    import dist_utils as du
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    du.init_from_env("gloo")
    g = torch.Generator().manual_seed(5)
    Bg, D, K, C = 12, 10, 8, 4
    x_img, x_txt = torch.randn(Bg, D, generator=g), torch.randn(Bg, D, generator=g)
    lab = (torch.rand(Bg, C, generator=g) < 0.4).float()

    def build():
        torch.manual_seed(11)
        return torch.nn.Linear(D, K), torch.nn.Linear(D, K), torch.nn.Parameter(torch.randn(C, K))
    # one process, whole batch
    fi, ft, px = build()
    ref = _pair_loss(torch.tanh(fi(x_img)), torch.tanh(ft(x_txt)), lab, px)
    ref.backward()
    # two ranks, half a batch each, loss on the gathered rows, gradient means over the ranks
    gi, gt, gp = build()
    n = Bg // world
    sl = slice(rank * n, (rank + 1) * n)
    sync = du.GradSync([list(gi.parameters()), list(gt.parameters()), [gp]])
    hi, ht, lb = du.gather_loss_inputs(torch.tanh(gi(x_img[sl])), torch.tanh(gt(x_txt[sl])), lab[sl])
    assert hi.shape == (Bg, K) and torch.equal(lb, lab)
    loss = _pair_loss(hi, ht, lb, gp)
    loss.backward()
    sync.finish()
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref)), (float(loss), float(ref))
    for a, b in zip(list(gi.parameters()) + list(gt.parameters()) + [gp], list(fi.parameters()) + list(ft.parameters()) + [px]):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6 * float(b.grad.abs().max())), (a.grad - b.grad).abs().max()
    # without gradients the same call is a plain fused all-gather
    with torch.no_grad():
        h2, = du.gather_loss_inputs(torch.tanh(gi(x_img[sl])))
    assert torch.allclose(h2, torch.tanh(gi(x_img)), atol=1e-7)
    # a draw shared by the ranks
    t = torch.full((3,), float(rank + 7))
    assert torch.equal(du.broadcast_tensor_(t, 0), torch.full((3,), 7.0))
    assert du.query_shard(11) == du.shard_range(11, rank, world)
    sync.remove()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    open(os.path.join(out_dir, f"gl{rank}"), "w").write("ok")


def _bucket_worker(rank, world, port, out_dir):
    """GradSync's in-place bucket route on gloo: a stand-in for a native tower (an autograd Function whose backward writes all
    gradients into ONE flat buffer from model/base/train_ops.py and reports every finished part to BUCKET_SINK) must end with the
    means of allreduce_mean_, with the parameters' .grad being views of the flat buffer (no packing, no copy-back)."""
    from conftest import PKG  # noqa: F401
    import dist_utils as du
    from model.base import train_ops as T
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    du.init_from_env("gloo")
    layers, nhead = 5, 3
    torch.manual_seed(2)
    params = [torch.nn.Parameter(torch.randn(4 * (i % 3 + 1), 4)) for i in range(nhead + 12 * layers)]
    seen = {}

    class FakeTower(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, *ps):
            ctx.ps = ps
            return x.sum() * sum(p.sum() for p in ps)

        @staticmethod
        def backward(ctx, g):
            ranges = T._layer_parts(layers)
            order, bounds = T._part_order(nhead, (1, 2), layers, ranges)
            grads, flat = T._grad_buffers(list(ctx.ps), order)
            seen["flat"], seen["ranges"] = flat, ranges
            done = set()
            for k, (hi, lo) in enumerate(ranges):          # what cmh_*_backward_part(hi, lo) writes
                idx = ([1, 2] if k == 0 else []) + [nhead + 12 * l + j for l in range(lo, hi) for j in range(12)] + \
                      ([0] if k == len(ranges) - 1 else [])
                for i in idx:
                    grads[i].fill_(float((rank + 1) * (i + 1)))
                done |= set(idx)
                T._sink(flat, list(ctx.ps), grads, order, bounds, k)
            assert done == set(range(len(ctx.ps)))
            return (None,) + tuple(grads)

    head = torch.nn.Linear(3, 2)
    sync = du.GradSync([params, list(head.parameters())])
    for step in range(2):
        for p in params + list(head.parameters()):
            p.grad = None
        (FakeTower.apply(torch.ones(2), *params) + head(torch.ones(3) * (rank + 1)).sum()).backward()
        sync.finish()
        assert seen["ranges"] == [(5, 4), (4, 2), (2, 0)] and [n for _, n in sync.bucket_log] == [2 + 12, 24, 24 + 1], sync.bucket_log
        flat = seen["flat"]
        for i, p in enumerate(params):
            assert torch.equal(p.grad, torch.full_like(p, 1.5 * (i + 1))), i                            # mean of (i+1) and 2 (i+1)
            assert flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4, "gradient was copied out of the flat buffer"
        assert torch.equal(head.weight.grad, torch.full_like(head.weight, 1.5))                         # the hook route, same step
    sync.remove()
    assert T.BUCKET_SINK is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    open(os.path.join(out_dir, f"bk{rank}"), "w").write("ok")


def test_world2_in_place_gradient_buckets(tmp_path):
    port = _free_port()
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "bk0").exists() and (tmp_path / "bk1").exists()


def test_world2_global_batch_loss_equals_single_process(tmp_path):
    """gather_loss_inputs + GradSync: loss and every gradient of a 2-rank step equal a 1-process step on the concatenated batch."""
    port = _free_port()
    mp.spawn(_global_loss_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "gl0").exists() and (tmp_path / "gl1").exists()


def test_world2_gloo_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_range_covers_everything():
    from conftest import PKG  # noqa: F401
    import dist_utils as du
    for n in (0, 1, 7, 256, 5000, 190834):
        for world in (1, 2, 3, 4, 8):
            spans = [du.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
