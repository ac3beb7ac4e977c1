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
    assert lib.cmh_version() == N.ABI_VERSION == 6
    assert lib.cmh_last_error() is not None


def test_host_side_argument_errors_without_gpu():
    """Pure host-side validation paths of the ABI (no kernel is launched)."""
    import cmh_native as N
    lib = N.lib()
    assert lib.cmh_map_workspace_bytes(0, 10, 16, 0) == 0
    assert lib.cmh_map_workspace_bytes(100, 1000, 64, 0) == 4096 + 64      # fits LDS (stamps area + the "no zeros in the database codes" word)
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


def test_argument_validation_of_the_later_entry_points():
    """Null pointers and impossible sizes come back as negative status codes with a message — nothing is launched, so this runs
    without a GPU (input pipeline, optimiser, backward building blocks, MITH training entry points, host tokenizer)."""
    import ctypes as C
    import cmh_native as N
    lib = N.lib()
    three = (C.c_float * 3)(0, 0, 0)
    calls = [
        lambda: lib.cmh_image_preprocess(None, None, None, 4, 10, 10, 16, 1, three, three, None, None, None, 0, None),
        lambda: lib.cmh_image_normalize(None, None, 4, 16, three, three, None, None),
        lambda: lib.cmh_bert_adam_step(None, 0, 0.9, 0.98, 1e-6, None, 0, None),
        lambda: lib.cmh_transpose(None, 0, None, 0, 4, 4, None),
        lambda: lib.cmh_layernorm_backward(None, 0, None, 0, None, 4, 8, None, 0, None, None, None, 0, None),
        lambda: lib.cmh_attention_backward(1, None, None, None, None, 1, 8, 64, 0, None, None),
        lambda: lib.cmh_linear_wgrad(1, None, 0, None, 0, 64, 64, 64, None, None, None, 0, None),
        lambda: lib.cmh_dnph_loss_backward(*([None] * 8), 4, 16, 8, 1.0, 0.1, None, *([None] * 5), None, 0, None),
        lambda: lib.cmh_twdh_loss_backward(None, None, None, 4, 8, None, None, None, None, None),
        lambda: lib.cmh_batchnorm1d_backward(None, None, 1e-5, None, None, None, None, 4, 8, None),
        lambda: lib.cmh_mith_lta_backward(None, None, None, None, 2, 8, 0, 8, 16, 64, 8, None),
        lambda: lib.cmh_blocks_forward_train(None, 2, 0, None, None, 2, 8, 128, None, 0, None),
        lambda: lib.cmh_blocks_backward(None, None, 2, 0, None, None, 2, 8, 128, None, 0, None),
        lambda: lib.cmh_mith_bayesian_loss_backward(None, None, None, None, 10, 4, 16, 8, None, None, None, 0, None),
        lambda: lib.cmh_info_nce_backward(None, None, 8, 4, 64, 0.07, None, None, None, None, 0, None),
        lambda: lib.cmh_sq_diff_sum_backward(None, None, 16, None, None, None, None),
        lambda: lib.cmh_gelu(None, None, 16, None),
        lambda: lib.cmh_l2_normalize_backward(None, None, None, 4, 8, None),
        lambda: lib.cmh_bitwise_hash_backward(*([None] * 7), 2, 4, 8, None),
        lambda: lib.cmh_vit_forward_train_tokens(None, None, 2, None, None, 0, None),
        lambda: lib.cmh_text_backward_tokens(None, None, 2, 8, None, None, None, None, 0, None),
        lambda: lib.cmh_bpe_create(None, 0, None),
        lambda: lib.cmh_bpe_encode_captions(None, None, None, 1, 8, None, None, 1),
    ]
    for k, call in enumerate(calls):
        rc = call()
        assert rc < 0, (k, rc)
        assert len(lib.cmh_last_error()) > 0
    assert lib.cmh_image_preprocess_workspace_bytes(0, 10, 10, 16) == 0 and lib.cmh_blocks_train_bytes(0, 0, 8, 128, 2) == 0
    assert lib.cmh_bpe_vocab_size(None) == 0


def test_struct_layouts_of_the_header_match_the_ctypes_mirrors(tmp_path):
    """The host binding re-declares the ABI's structs in ctypes; a field added on one side only would shift every pointer behind it
    (the version check catches a stale LIBRARY, not a stale mirror).  gcc compiles include/cmh.h and reports sizeof / the offset of
    each struct's last member; the ctypes classes must agree."""
    import ctypes as C
    import cmh_native as N
    pairs = {"cmh_block_weights": (N.BlockWeights, "act_scale"), "cmh_vit_weights": (N.VitWeights, "blocks"),
             "cmh_text_weights": (N.TextWeights, "blocks"), "cmh_taps": (N.Taps, "count"),
             "cmh_block_grads": (N.BlockGrads, "proj_b"), "cmh_vit_grads": (N.VitGrads, None), "cmh_text_grads": (N.TextGrads, None),
             "cmh_adam_tensor": (N.AdamTensor, None)}
    pairs = {k: v for k, v in pairs.items() if v[0] is not None}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "cmh.h"', 'int main(void) {']
    for name, (_, last) in pairs.items():
        src.append(f'  printf("{name} %zu %zu\\n", sizeof({name}), {f"offsetof({name}, {last})" if last else "(size_t)0"});')
    src += ['  return 0;', '}']
    c_file, exe = tmp_path / "layout.c", tmp_path / "layout"
    c_file.write_text("\n".join(src))
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c_file), "-o", str(exe)])
    got = {ln.split()[0]: (int(ln.split()[1]), int(ln.split()[2])) for ln in subprocess.check_output([str(exe)], text=True).splitlines()}
    for name, (cls, last) in pairs.items():
        assert C.sizeof(cls) == got[name][0], (name, C.sizeof(cls), got[name][0])
        if last:
            assert getattr(cls, last).offset == got[name][1], (name, last)
