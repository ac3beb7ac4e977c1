"""The trainers with one process per GPU: two ranks (gloo, sharing the one GPU of the test box) run main.py's flow —
sharded batches, loss-side all-gather, gradient means queued inside backward, sharded evaluation gathered before ranking.
Both ranks must end with identical weights and identical mAPs; only rank 0 writes checkpoints and .mat files."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("method", ["DSPH", "DCHMT", "TwDH", "DNPH", "MITH", "DNpH", "DMsH_LN", "DHaPH"])
def test_two_ranks_stay_replicas(tmp_path, method):
    env = dict(os.environ, CMH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "two_rank_driver.py"), str(tmp_path), method]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, "\n".join(l for l in res.stderr.splitlines() if l.startswith("[rank0]"))[-3000:]
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1))
    assert r0["main"] and not r1["main"]
    assert r0["checksum"] == r1["checksum"], "replicas diverged"
    assert r0["maps"] == r1["maps"]
    assert r0["batches"] == r1["batches"] == 2                               # 50 items -> 25 per rank -> 2 batches of <= 16
    if method == "DSPH":
        assert r0["steps"] == 4
    run = tmp_path / "run" / method / "synthetic" / "16"
    assert (run / "model-1.pth").exists() and (run / "train.log").exists() and (run / "train.rank1.log").exists()
    assert len(list(run.glob("model-*.pth"))) == 2


@pytest.mark.parametrize("method", ["DSPH", "DCHMT", "MITH", "DNpH", "DMsH_LN", "DHaPH"])
def test_two_ranks_equal_one_rank(tmp_path, method):
    """north_star / SURVEY 8e: the pairwise loss sees the GLOBAL batch and evaluation shards the queries.  One step of two ranks
    (8 samples each) must reproduce one step of one process on the 16 samples: the same loss, the same gradients on every
    replica, and bit-identical mAPs from the query-sharded evaluation.  (DNPH's Hungarian assignment and TwDH's BatchNorm1d are
    rank-local by construction, DESIGN 6, so they are covered by test_two_ranks_stay_replicas only.)"""
    import numpy as np
    env = dict(os.environ, CMH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    drv = os.path.join(HERE, "two_rank_equiv_driver.py")
    one = subprocess.run([sys.executable, drv, str(tmp_path), method], env={k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE")},
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), drv, str(tmp_path), method], env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, "\n".join(l for l in two.stderr.splitlines() if "Error" in l or l.startswith("[rank0]"))[-3000:]
    ref = json.load(open(tmp_path / "res_w1r0.json"))
    g_ref = np.load(tmp_path / "grads_w1r0.npz")
    assert ref["n_grads"] >= 8
    for r in (0, 1):
        got = json.load(open(tmp_path / f"res_w2r{r}.json"))
        if method != "MITH":     # MITH logs a rank-local mean for its per-sample terms; its gradients are the global ones
            assert abs(got["loss"] - ref["loss"]) <= 2e-6 * abs(ref["loss"]), (got["loss"], ref["loss"])
        assert got["maps"] == ref["maps"], (got["maps"], ref["maps"])
        # the towers' gradients travelled as in-place buckets of the flat buffer (2 towers x >= 2 parts; MITH's all-token trunk: one
        # bucket per tower) and the parameters' .grad are views of that buffer: nothing packed, nothing copied back
        assert got["buckets"] is not None and len(got["buckets"]) >= (2 if method == "MITH" else 4), got["buckets"]
        assert got["tower_grad_is_view"]
        g = np.load(tmp_path / f"grads_w2r{r}.npz")
        assert set(g.files) == set(g_ref.files)
        for name in g_ref.files:
            scale = float(np.abs(g_ref[name]).max())
            tol = (2e-4 if method == "MITH" else 2e-5) * scale + 1e-12
            assert float(np.abs(g[name] - g_ref[name]).max()) <= tol, (name, float(np.abs(g[name] - g_ref[name]).max()), scale)


def test_bench_self_launch_two_ranks_end_to_end():
    """`python bench.py --gpus 2` exactly as a user (or the driver's 8-GPU node) starts it bare: the launcher parent never touches the
    GPU, starts two fresh ranks under torch.distributed.run and passes rank 0's JSON line through.  Here both ranks share the test
    box's one GPU and the collectives run on gloo (CMH_DIST_BACKEND) - what is proved is the launcher, the rendezvous, the N > 1
    branch of the step (fused all-gather -> global-batch loss), the max-over-ranks timing and the shape of the line; RCCL itself is
    covered by tests/test_gpu_rccl_one_rank.py."""
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CMH_FORCE_DIST")}
    env.update(CMH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--batch", "32",
           "--no-cpu-baseline", "--no-dense-text", "--map-queries", "64", "--map-db", "500", "--train-step"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["collectives"] == "gloo" and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 64 and rec["config"]["per_gpu_batch"] == 32
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and abs(rec["per_gpu_value"] * 2 - rec["value"]) < 1e-6 * rec["value"] + 0.02
    assert "relaunched" not in rec
    assert "error" not in rec["train_step"] and rec["train_step"]["loss"] == rec["train_step"]["loss"]      # finite: the N > 1 training step ran
    assert rec["map_eval"]["mAP_i2t"] > 0
