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


@pytest.mark.parametrize("method", ["DSPH", "DCHMT", "TwDH", "DNPH", "MITH"])
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
