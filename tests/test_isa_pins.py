"""The hot GEMM kernels' register allocation and instruction stream are pinned (tools/isa_pin.py, tests/golden/isa_pins.json).

The 160-row instantiations of gemm_wide_kernel sit exactly at 256 VGPRs with a few bytes of scratch outside the K loop; an edit
anywhere in that translation unit can move the allocator (round 3: -0.6 % on the headline from dead code).  A deliberate kernel
change re-records the pins in the same commit (python3 tools/isa_pin.py record) next to a bench A/B; an accidental one fails here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_default_gemm_kernels_match_their_pinned_isa():
    import isa_pin
    pinned = json.load(open(isa_pin.PINS))
    cur = isa_pin.current()
    want = pinned["kernels"]
    assert sorted(cur) == sorted(want), (sorted(set(cur) ^ set(want)))
    for name, rec in want.items():
        got = cur[name]
        # the budget first (readable failure), then the stream itself
        assert (got["vgpr"], got["scratch"], got["lds"]) == (rec["vgpr"], rec["scratch"], rec["lds"]), (name, rec, got)
        assert got["sha256"] == rec["sha256"] and got["instructions"] == rec["instructions"], (name, rec, got)
    # the invariants the design relies on, stated once more in plain numbers: every variant a launch can select fits two waves per
    # SIMD (<= 256 registers) and one workgroup's LDS fits the CU (<= 160 KiB)
    for name, rec in cur.items():
        assert rec["vgpr"] <= 256 and rec["lds"] <= 160 * 1024, (name, rec)


def test_no_experiment_symbols_in_the_default_library():
    """Round 3's measured-slower experiments (256 x 256 'big' tile, LayerNorm fold) and round 1's register-staged kernel are not part
    of libcmh.so."""
    import subprocess
    lib = os.path.join(ROOT, "clip-based-cross-modal-hashing_amd", "csrc", "build", "libcmh.so")
    syms = subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True)
    for bad in ("gemm_big", "ln_fold", "lnfold", "set_gemm_big"):
        assert bad not in syms, bad
