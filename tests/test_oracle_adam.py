"""oracle/adam_oracle.py (numpy restatement of the reference's BertAdam step) against tests/golden/adam.npz, which the
reference's own BertAdam produced (tests/golden/make_golden4.py)."""
import numpy as np
import pytest

import adamutil as au


@pytest.mark.parametrize("tag", ["trainer", "plain"])
def test_oracle_matches_reference_bertadam(golden, tag):
    g = golden("adam.npz")
    ps, gs, ms, vs = au.run_oracle(tag)
    for i in range(int(g["nshapes"])):
        for name, arr in (("p", ps[i]), ("m", ms[i]), ("v", vs[i]), ("g", gs[i])):
            ref = g[f"{tag}_{name}{i}"]
            # the only freedom left is ATen's vectorised summation order inside the gradient norm: the clip coefficient, hence
            # every clipped gradient, may move by 1 ulp -> a few ulps of the array's largest magnitude after 4 steps
            np.testing.assert_allclose(au.cut(arr), ref, rtol=3e-7, atol=au.atol(ref), err_msg=f"{tag} {name}{i}")


def test_schedules_match_reference_lr_trace(golden):
    import oracle.adam_oracle as ao
    g = golden("adam.npz")
    kw = au.CONFIGS["trainer"]
    for s in range(int(g["steps"])):
        # get_lr() after step s reports the rate of step s+1 (state['step'] was incremented, optimization.py:166)
        want = g[f"trainer_lr_step{s}"]
        x = (s + 1) / kw["t_total"]
        assert abs(want[-1] - kw["lr"] * ao.schedule(kw["schedule"], x, kw["warmup"])) < 1e-15
        assert abs(want[0] - au.GROUP0_LR * ao.schedule(kw["schedule"], x, kw["warmup"])) < 1e-15
