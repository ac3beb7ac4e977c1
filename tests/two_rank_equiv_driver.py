"""Child of tests/test_gpu_two_ranks.py::test_two_ranks_equal_one_rank: ONE optimisation step of a trainer on a fixed global batch
of 16 samples, either alone (world 1, all 16 rows) or as one of two ranks (8 rows each, gloo, both on cuda:0), then one
evaluation.  Leaves the loss, sampled gradients and the four mAPs for the parent to compare."""
import argparse, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", "clip-based-cross-modal-hashing_amd"), os.path.join(HERE, "golden"), HERE]
import numpy as np
import torch
import recipe
import dist_utils as du
import dataset.synthetic as ds
import main

out, method = sys.argv[1], sys.argv[2]
rank, world, _ = du.init_from_env()
torch.cuda.set_device(0)
tag = f"w{world}r{rank}" + ("f" if du.forced() else "")     # f: a FORCED group of one rank (CMH_FORCE_DIST=1: every collective runs)
ck = os.path.join(out, f"clip_{tag}.pt")
torch.save({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(dict(recipe.CLIP_TINY, embed_dim=512), 7).items()}, ck)
ds.SOT, ds.EOT = 510, 511
sys.argv = ["main.py", "-clip-path", ck, "--save-dir", os.path.join(out, f"run_{tag}"), "--batch-size", "16", "--num-workers", "0",
            "--resolution", "64", "--max-words", "16", "--query-num", "25", "--train-num", "50", "--synthetic-size", "120",
            "--gemm-dtype", "f32", "--epochs", "0"]
torch.manual_seed(100)                                       # the same heads / loss parameters in every process
tr = main.trainers[method](argparse.Namespace(method=method, dataset="synthetic", output_dim=16, is_train=True), 0)
tr.model.eval()                                              # heads without dropout; autograd stays on
data = tr.train_loader.dataset
items = [data[i] for i in range(16)]
lo, hi = du.shard_range(16, rank, world)
cols = list(zip(*items[lo:hi]))
batch = [torch.stack([torch.as_tensor(v) for v in c]) for c in cols]
for name in ("optimizer", "optimizer_loss"):                 # keep the weights: this test compares gradients
    if hasattr(tr, name):
        getattr(tr, name).step = lambda *a, **k: None
if method == "DMsH_LN":
    # a freshly initialised LabelNet marks every pair as similar - the loss is the constant zero and there is nothing to compare;
    # centre its codes on the 16 items (every rank computes the same bias from the same items)
    with torch.no_grad():
        lab16 = torch.stack([torch.as_tensor(it[2]) for it in items]).to(0).float()
        feat = torch.relu(lab16 @ tr.L_net.fc1.weight.t() + tr.L_net.fc1.bias)
        tr.L_net.fc2.bias.copy_(-(feat.mean(0) @ tr.L_net.fc2.weight.t()))
if method == "MITH":
    image, text, kpm, label, index = batch
    tr.change_state(mode="valid")
    od = tr.model(image.to(0), text.to(0), kpm.to(0))
    losses = tr.compute_loss(od, label)
    loss = sum(losses.values())
    tr.optimizer.zero_grad()
    tr.backward(loss)
else:
    image, text, label, index = batch
    loss = tr._step(image, text, label) if method in ("DSPH", "DNPH", "DCHMT", "DNpH", "DMsH_LN", "DHaPH") else tr._step(image, text, label, index)
grads = {}
for name, p in tr.model.named_parameters():
    if p.grad is not None and any(k in name for k in ("proj", "hash", "resblocks.0.attn.in_proj_weight", "resblocks.1.mlp.c_fc.weight",
                                                       "positional_embedding", "ln_final", "ln_post", "conv1")):
        grads[name] = p.grad.detach().float().cpu().numpy()
for mod in ("hyp",):
    if hasattr(tr, mod):
        for name, p in getattr(tr, mod).named_parameters():
            grads[f"{mod}.{name}"] = p.grad.detach().float().cpu().numpy()
tr.change_state(mode="valid")
r = tr.valid(0)
maps = [float(v) for v in (r["long"] if isinstance(r, dict) else r)]
np.savez(os.path.join(out, f"grads_{tag}.npz"), **grads)
sync = getattr(tr, "_grad_sync", None)
tower_p = tr.model.clip.visual.transformer.resblocks[0].attn.in_proj_weight
json.dump({"loss": float(loss.detach()), "maps": maps, "n_grads": len(grads),
           "backend": torch.distributed.get_backend() if du.active() else None,
           "buckets": [list(b) for b in sync.bucket_log] if sync is not None else None,          # in-place messages of the step
           "tower_grad_is_view": tower_p.grad is not None and tower_p.grad.untyped_storage().nbytes() > 4 * tower_p.grad.numel()},
          open(os.path.join(out, f"res_{tag}.json"), "w"))
if du.active():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
