"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).

CPU restatement of the reference's hot path, pinned to outputs of the reference itself
(tests/golden/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; the product path (clip-based-cross-modal-hashing_amd/) never does.
"""
from . import clip_oracle  # noqa: F401
from .map_oracle import hamming_row, map_k, sort_perm, sort_perm_depth, build  # noqa: F401
