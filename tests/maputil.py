"""Rebuild the mAP fixture inputs from the recipe (tests/golden/make_golden.py::gen_map)."""
import numpy as np

import recipe

CASES = ["rand_1k_16", "rand_1k_64", "corr_1k_64", "corr_1k_64_k50", "zeros_300_32", "tiny_17_16",
         "tiny_16_8", "odd_5003_128", "flickr_20015_64", "nus_190k_128", "coco_117k_64"]


def case_inputs(g, name):
    Q, N, K, C, k = (int(v) for v in g[f"{name}_shape"])
    kind, zeros, p = str(g[f"{name}_kind"]), int(g[f"{name}_zeros"]), float(g[f"{name}_p"])
    seed = 1234
    qL = recipe.labels(Q, C, seed, p=p, tag=f"map_qL_{name}")
    rL = recipe.labels(N, C, seed, p=p, tag=f"map_rL_{name}")
    if kind == "rand":
        qB = recipe.sign_codes(Q, K, seed, f"map_qB_{name}", zeros=zeros)
        rB = recipe.sign_codes(N, K, seed, f"map_rB_{name}", zeros=zeros * 3)
    else:
        qB = recipe.correlated_codes(qL, K, seed, f"map_qB_{name}")
        rB = recipe.correlated_codes(rL, K, seed, f"map_rB_{name}")
    return qB, rB, qL, rL, (None if k < 0 else k)
