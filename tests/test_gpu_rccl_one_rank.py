"""RCCL executed from this repository on the one GPU a test box has.

Every other distributed test runs on gloo (two ranks sharing the GPU): the `nccl` branch of dist_utils - the communicator bound to the
device, all_gather_into_tensor / all_reduce / broadcast on DEVICE tensors, GradSync's asynchronous in-place buckets and the stream
hand-over behind them - would otherwise meet RCCL for the first time on the driver's 8-GPU node.  CMH_FORCE_DIST=1 makes a process
join a group of ONE rank and run every collective of the path anyway (dist_utils.forced / active).  A group of one cannot show a ring
over xGMI or a rank-dependent indexing bug (the gloo tests cover those); it does show that every call site hands RCCL tensors, dtypes,
devices and streams it accepts, and that the results are the plain single-process results BIT FOR BIT (a sum over one rank, a
division by one).  Call sites served: train/MITH/hash_train.py:72-78, train/base.py:130-148, 259-262 of the reference."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CMH_DIST_BACKEND", "CMH_FORCE_DIST")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), **kw)
    return env


@pytest.mark.parametrize("method", ["DSPH", "MITH"])
def test_one_rank_on_rccl_reproduces_the_plain_step(tmp_path, method):
    """One optimisation step + one evaluation of a trainer, (a) with no process group and (b) in a forced `nccl` group of one:
    broadcast of the initial weights, the fused differentiable all-gather of the loss inputs, the towers' gradients as in-place
    buckets all-reduced from inside the backward pass, the packed code gather and the query-sharded AP gather all run on RCCL -
    and leave the same loss, the same gradients and the same mAPs, bit for bit."""
    drv = os.path.join(HERE, "two_rank_equiv_driver.py")
    plain = subprocess.run([sys.executable, drv, str(tmp_path), method], env=_env(), capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-3000:]
    forced = subprocess.run([sys.executable, drv, str(tmp_path), method], env=_env(CMH_FORCE_DIST="1", CMH_DIST_BACKEND="nccl"),
                            capture_output=True, text=True, timeout=600)
    assert forced.returncode == 0, forced.stderr[-3000:]
    ref, got = json.load(open(tmp_path / "res_w1r0.json")), json.load(open(tmp_path / "res_w1r0f.json"))
    assert ref["backend"] is None and got["backend"] == "nccl"
    assert got["loss"] == ref["loss"] and got["maps"] == ref["maps"], (got, ref)
    assert ref["buckets"] is None
    assert got["buckets"] is not None and len(got["buckets"]) >= (2 if method == "MITH" else 4), got["buckets"]
    assert got["tower_grad_is_view"]
    g_ref, g = np.load(tmp_path / "grads_w1r0.npz"), np.load(tmp_path / "grads_w1r0f.npz")
    assert set(g.files) == set(g_ref.files) and len(g.files) >= 8
    for name in g_ref.files:
        assert np.array_equal(g[name], g_ref[name]), name


def test_collectives_of_the_path_on_rccl(tmp_path):
    """dist_utils' primitives one by one on device tensors in a forced nccl group of one (tests/rccl_one_rank_driver.py)."""
    res = subprocess.run([sys.executable, os.path.join(HERE, "rccl_one_rank_driver.py"), str(tmp_path / "out.json")],
                         env=_env(CMH_FORCE_DIST="1", CMH_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.load(open(tmp_path / "out.json"))
    assert out["backend"] == "nccl" and out["world"] == 1
    assert all(out["checks"].values()), out["checks"]
    assert len(out["checks"]) >= 9


def test_bench_multi_gpu_branch_on_rccl(tmp_path):
    """bench.py's N > 1 branch (fused code all-gather inside the step, MAX all-reduce of the elapsed time, query-sharded mAP leg, the
    data-parallel training step through gather_loss_inputs + GradSync) on RCCL in a group of one: the line the driver will ask for
    on 8 GPUs comes out whole."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--train-step", "--no-cpu-baseline",
           "--no-dense-text", "--no-input-pipeline", "--no-config-legs", "--no-precision-legs", "--no-towers-ab"]
    res = subprocess.run(cmd, env=_env(CMH_FORCE_DIST="1", CMH_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line.get("collectives") == "nccl (forced group of one)", line.get("collectives")
    assert "error" not in line["train_step"] and line["train_step"]["ms"] > 0, line["train_step"]
    assert "error" not in line.get("map_eval", {}), line.get("map_eval")
