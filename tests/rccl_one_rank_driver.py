"""Child of tests/test_gpu_rccl_one_rank.py: every collective dist_utils issues, on device tensors, in a forced nccl group of one."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "clip-based-cross-modal-hashing_amd")]
import torch
import dist_utils as du

rank, world, local = du.init_from_env()
assert du.forced() and du.active() and world == 1
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
checks = {}

# all_gather_rows: equal blocks and the ragged (padded) form
blk = torch.randn(37, 152, generator=g).to(dev)
checks["all_gather_rows"] = torch.equal(du.all_gather_rows(blk), blk)
checks["all_gather_rows_ragged"] = torch.equal(du.all_gather_rows(blk, du.row_counts(37, dev)), blk)
checks["row_counts"] = du.row_counts(37, dev) == [37]
# int32 blocks travel too (the packed code gather of train/base.py::_gather_code_shards)
codes = torch.randint(-2 ** 31, 2 ** 31 - 1, (50, 5), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
checks["all_gather_rows_int32"] = torch.equal(du.all_gather_rows(codes, du.row_counts(50, dev)), codes)

# the exchange step of a training iteration, forward and backward
h = torch.randn(16, 64, generator=g).to(dev).requires_grad_(True)
lab = (torch.rand(16, 24, generator=g) < 0.2).float().to(dev)
hg, lg = du.gather_loss_inputs(h, lab)
checks["gather_loss_inputs_forward"] = torch.equal(hg, h.detach()) and torch.equal(lg, lab)
w = torch.randn(16, 64, generator=g).to(dev)
(hg * w).sum().backward()
checks["gather_loss_inputs_backward"] = torch.equal(h.grad, w)        # rows of this rank, times world = 1

# evaluation: per-query APs gathered in query order, summed by the ranking kernel's own mean
ap = torch.rand(101, generator=g).to(dev)
full = du.gather_query_sharded_ap(ap, 101)
import cmh_native as N
checks["gather_query_sharded_ap"] = torch.equal(full, ap)
checks["mean_in_query_order"] = float(du.mean_in_query_order(full)) == float(N.map_mean(ap))

# replicas: broadcast of modules, of a tensor; bucketed gradient means
lin = torch.nn.Linear(8, 4).to(dev)
before = [p.detach().clone() for p in lin.parameters()]
du.broadcast_modules_([lin])
checks["broadcast_modules"] = all(torch.equal(a, b) for a, b in zip(before, lin.parameters()))
t = torch.randn(3, 3, generator=g).to(dev)
checks["broadcast_tensor"] = torch.equal(du.broadcast_tensor_(t.clone()), t)
grads = [torch.randn(n, generator=g).to(dev) for n in (1000, 70000, 3, 512 * 512)]
want = [x.clone() for x in grads]
du.allreduce_mean_(grads, bucket_bytes=256 << 10)                       # several buckets in flight
checks["allreduce_mean_buckets"] = all(torch.equal(a, b) for a, b in zip(grads, want))

# GradSync's hook route on plain modules (the towers' in-place buckets are covered by the trainer step of the same test file)
net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2)).to(dev)
sync = du.GradSync([list(net.parameters())])
x = torch.randn(5, 8, generator=g).to(dev)
for _ in range(2):                                                      # the second step sends from inside backward (groups learnt)
    for p in net.parameters():
        p.grad = None
    net(x).square().sum().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    sync.finish()
    ok = all(torch.equal(p.grad, r) for p, r in zip(net.parameters(), ref))
checks["gradsync_hooks"] = ok
sync.remove()

torch.cuda.synchronize()
json.dump({"backend": torch.distributed.get_backend(), "world": world, "checks": checks}, open(sys.argv[1], "w"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
