"""CPU checks of the C-ABI boundary: libcmh.so builds/loads, exports every symbol include/cmh.h declares
(and the ctypes table binds exactly those), and the product never imports the oracle."""
import os
import re
import subprocess

import pytest

from conftest import PKG, ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "cmh.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cmh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import cmh_native as N
    lib = N.lib()                                   # raises if the .so is missing
    declared = _declared()
    assert len(declared) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", N.LIB_PATH], text=True)
    exported = set(re.findall(r" T (cmh_[a-z0-9_]+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(N.SIGNATURES) == set(declared)
    assert lib.cmh_version() == 1
    assert lib.cmh_last_error() is not None


def test_host_side_argument_errors_without_gpu():
    """Pure host-side validation paths of the ABI (no kernel is launched)."""
    import cmh_native as N
    lib = N.lib()
    assert lib.cmh_map_workspace_bytes(0, 10, 16, 0) == 0
    assert lib.cmh_map_workspace_bytes(100, 1000, 64, 0) == 256            # fits LDS
    assert lib.cmh_map_workspace_bytes(100, 190000, 128, 0) > 100 * 190000  # global slices
    assert lib.cmh_loss_workspace_bytes(256, 64, 24) > 4 * 256 * 64 * 4
    rc = lib.cmh_pack_codes(None, 10, 16, None, None, None, None)
    assert rc == -1 and b"null" in lib.cmh_last_error()
    rc = lib.cmh_linear_gemm(0, 1, 1, None, None, 1, 8, 100, 64, 0, None)   # N not a multiple of 128
    assert rc == -1 and b"multiple" in lib.cmh_last_error()


def test_cpu_tensors_are_refused():
    import torch
    import cmh_native as N
    with pytest.raises(N.NativeError):
        N.sign_codes(torch.zeros(4, 4))


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
