"""Grouped launches (round 4): layer i of the image and the text tower as ONE launch of the wide GEMM kernel
(csrc/gemm_wide.hip, template parameter GRP; include/cmh.h: cmh_linear_gemm_grouped, cmh_clip_encode_pair).

The contract is bit-identity: a grouped launch walks the same tiles with the same K order and the same epilogue as the two plain
launches it replaces (reference: two independent nn.Linear calls, model/base/model.py:171-196, one per tower), so every comparison
below is torch.equal - all three arithmetic modes, packed and dense captions, every tile height, device-side row counts."""
import pytest
import torch

import recipe

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(shape, g, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


# the four GEMMs of a block of both towers at batch 256 (packed text rows), plus shapes that stress the schedule: one problem with
# fewer tiles than an XCD has workgroups, ragged last tiles, K-steps 8 vs 48
SHAPES = [
    ((12800, 2304, 768), (10499, 1536, 512)),     # QKV
    ((12800, 768, 768), (10499, 512, 512)),       # out_proj (residual)
    ((12800, 3072, 768), (10499, 2048, 512)),     # c_fc (QuickGELU)
    ((12800, 768, 3072), (10499, 512, 2048)),     # c_proj (residual)
    ((2100, 256, 512), (4000, 1024, 256)),        # 'a' shorter in K than 'b': the launcher swaps them
    ((2049, 512, 1024), (2500, 256, 1024)),       # a handful of tiles each
    ((9000, 1024, 512), (4100, 512, 1024)),       # QuickGELU again (kind 2): ragged last tiles, 8 / 16 K-steps, several tiles per workgroup
                                                  # at 128 rows - the deferred-activation variant of the wide kernel across a problem switch
]


@pytest.mark.parametrize("tile_rows", [-1, 96, 128, 160])
@pytest.mark.parametrize("case", range(len(SHAPES)))
def test_grouped_bf16_gemm_gives_the_plain_launches_bits(case, tile_rows):
    import cmh_native as N
    (Ma, Na, Ka), (Mb, Nb, Kb) = SHAPES[case]
    g = torch.Generator().manual_seed(100 + case)
    kind = case % 4                                   # 0 plain bias, 1 / 3 fp16 residual stream, 2 QuickGELU
    probs = []
    for (M, Nn, K) in ((Ma, Na, Ka), (Mb, Nb, Kb)):
        p = {"x": _rand((M, K), g).bfloat16().to(DEV), "w": _rand((Nn, K), g, K ** -0.5).bfloat16().to(DEV), "bias": _rand((Nn,), g).to(DEV)}
        if kind in (1, 3):
            p["residual"] = _rand((M, Nn), g).half().to(DEV)
        probs.append(p)
    out = "f16" if kind in (1, 3) else "bf16"
    try:
        N.lib().cmh_gemm_tuning(tile_rows, -1)
        ref = [N.linear_gemm(p["x"], p["w"], bias=p["bias"], residual=p.get("residual"), quickgelu=kind == 2, out_bf16=out == "bf16",
                             out_f16=out == "f16") for p in probs]
        got = N.linear_gemm_grouped(probs, quickgelu=kind == 2, out=out)
        # a device-side row count for the second problem (packed captions): rows past it are never written
        md = torch.tensor([Mb - 37], dtype=torch.int32, device=DEV)
        got_md = N.linear_gemm_grouped(probs, quickgelu=kind == 2, out=out, m_dev=(None, md))
    finally:
        N.lib().cmh_gemm_tuning(-1, -1)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    assert torch.equal(got_md[0], ref[0]) and torch.equal(got_md[1][:Mb - 37], ref[1][:Mb - 37])


@pytest.mark.parametrize("case", [0, 1, 3])
def test_grouped_f32_gemm_gives_the_plain_launches_bits(case):
    import cmh_native as N
    (Ma, Na, Ka), (Mb, Nb, Kb) = SHAPES[case]
    Ma, Mb = Ma // 4, Mb // 4                         # (the f32 MFMA runs at 1/16 of the bf16 rate)
    g = torch.Generator().manual_seed(200 + case)
    probs = []
    for (M, Nn, K) in ((Ma, Na, Ka), (Mb, Nb, Kb)):
        p = {"x": _rand((M, K), g).to(DEV), "w": _rand((Nn, K), g, K ** -0.5).to(DEV), "bias": _rand((Nn,), g).to(DEV)}
        if case in (1, 3):
            p["residual"] = _rand((M, Nn), g).to(DEV)
        probs.append(p)
    ref = [N.linear_gemm(p["x"], p["w"], bias=p["bias"], residual=p.get("residual")) for p in probs]
    got = N.linear_gemm_grouped(probs)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)


@pytest.mark.parametrize("out", ["bf16", "f16", "fp8"])
def test_grouped_fp8_gemm_gives_the_plain_launches_bits(out):
    import cmh_native as N
    (Ma, Na, Ka), (Mb, Nb, Kb) = SHAPES[{"bf16": 0, "f16": 1, "fp8": 2}[out]]
    g = torch.Generator().manual_seed(300)
    probs = []
    for (M, Nn, K) in ((Ma, Na, Ka), (Mb, Nb, Kb)):
        x8 = N.fp8_quantize(_rand((M, K), g).to(DEV), 1.0 / 64)
        w8, cs = N.fp8_quantize_weight(_rand((Nn, K), g, K ** -0.5).to(DEV))
        p = {"x": x8, "w": w8, "colscale": cs, "alpha": 1.0 / 64, "bias": _rand((Nn,), g).to(DEV), "out_scale": 3.0 if out == "fp8" else 1.0}
        if out == "f16":
            p["residual"] = _rand((M, Nn), g).half().to(DEV)
        probs.append(p)
    ref = [N.linear_gemm_fp8(p["x"], p["w"], p["colscale"], p["alpha"], bias=p["bias"], residual=p.get("residual"),
                             quickgelu=out == "fp8", out=out, out_scale=p["out_scale"]) for p in probs]
    got = N.linear_gemm_grouped(probs, quickgelu=out == "fp8", out=out)
    for r, o in zip(ref, got):
        assert torch.equal(r, o)


def test_grouping_can_be_switched_off_and_small_shapes_fall_back():
    """cmh_set_gemm_grouped(0) and shapes the wide kernel cannot take together (few rows, N % 256) run as two plain launches."""
    import cmh_native as N
    g = torch.Generator().manual_seed(5)
    probs = [{"x": _rand((300, 256), g).bfloat16().to(DEV), "w": _rand((384, 256), g, 0.06).bfloat16().to(DEV)},
             {"x": _rand((5000, 512), g).bfloat16().to(DEV), "w": _rand((512, 512), g, 0.04).bfloat16().to(DEV)}]
    ref = [N.linear_gemm(p["x"], p["w"], out_bf16=True) for p in probs]
    for on in (1, 0):
        try:
            N.set_gemm_grouped(on)
            got = N.linear_gemm_grouped(probs, out="bf16")
        finally:
            N.set_gemm_grouped(-1)
        assert all(torch.equal(r, o) for r, o in zip(ref, got))


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp8"])
@pytest.mark.parametrize("B", [8, 48, 256])       # 8: few rows - the GEMMs of both towers take the few-row kernel (no grouping)
def test_encode_pair_equals_the_two_single_tower_calls(mode, B):
    """cmh_clip_encode_pair (CLIP.encode_pair): ViT-B/32 at its real size, packed and dense captions - the features of both towers
    equal the separate encode_image / encode_text calls bit for bit; with grouping switched off the pair call still does."""
    import cmh_native as N
    from model.base.model import CLIP
    if mode == "f32" and B == 256:
        pytest.skip("the f32 mode at batch 256 adds 30 s for no new code path")
    cfg = recipe.CLIP_VITB32
    torch.manual_seed(21)
    m = CLIP(**cfg).to(DEV).float().set_gemm_dtype("bf16" if mode == "fp8" else mode)
    m.assume_frozen = True
    img = torch.from_numpy(recipe.images(B, cfg["image_resolution"], 5)).to(DEV)
    txt = torch.from_numpy(recipe.captions(B, 77, cfg["vocab_size"], 6)).to(DEV)
    if mode == "fp8":
        m.calibrate_fp8(img[:min(B, 16)], txt[:min(B, 16)])
        m.set_gemm_dtype("fp8")
    with torch.no_grad():
        for pack in (True, False):
            m.pack_text = pack
            ref = (m.encode_image(img).clone(), m.encode_text(txt).clone())
            got = tuple(t.clone() for t in m.encode_pair(img, txt))
            try:
                N.set_gemm_grouped(0)
                off = tuple(t.clone() for t in m.encode_pair(img, txt))
            finally:
                N.set_gemm_grouped(-1)
            for r, a, b in zip(ref, got, off):
                assert torch.isfinite(r).all() and r.abs().sum() > 0
                assert torch.equal(r, a) and torch.equal(r, b)


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_two_loader_batches_in_one_call_equal_two_calls(mode):
    """cmh_clip_encode_pair2 (CLIP.encode_pair2 / prefetch_pairs; the evaluation loop coalesces two loader batches, train/base.py
    ::_pipelined_batches): ViT-B/32, batches of 48 and 32 pairs as ONE run of the towers - every row's features equal the two
    separate encode_pair calls bit for bit, packed and dense captions, and prefetch_pairs hands each batch its own rows."""
    from model.base.model import CLIP
    cfg = recipe.CLIP_VITB32
    torch.manual_seed(23)
    m = CLIP(**cfg).to(DEV).float().set_gemm_dtype("bf16")
    m.assume_frozen = True
    ia, ib = (torch.from_numpy(recipe.images(n, cfg["image_resolution"], s)).to(DEV) for n, s in ((48, 5), (32, 6)))
    ta, tb = (torch.from_numpy(recipe.captions(n, 77, cfg["vocab_size"], s)).to(DEV) for n, s in ((48, 7), (32, 8)))
    if mode == "fp8":
        m.calibrate_fp8(ia[:16], ta[:16])
        m.set_gemm_dtype("fp8")
    with torch.no_grad():
        for pack in (True, False):
            m.pack_text = pack
            ra, rb = [t.clone() for t in m.encode_pair(ia, ta)], [t.clone() for t in m.encode_pair(ib, tb)]
            fi, ft = m.encode_pair2(ia, ta, ib, tb)
            assert torch.equal(fi[:48], ra[0]) and torch.equal(fi[48:], rb[0]) and torch.equal(ft[:48], ra[1]) and torch.equal(ft[48:], rb[1])
        m.prefetch_pairs([(ia, ta), (ib, tb)])
        assert torch.equal(m.encode_image(ib), rb[0]) and torch.equal(m.encode_text(ta), ra[1])      # any order, by tensor identity
        assert torch.equal(m.encode_image(ia), ra[0]) and torch.equal(m.encode_text(tb), rb[1]) and not m._pair_stash


def test_prefetch_pair_feeds_the_next_two_encodes():
    """CLIP.prefetch_pair (the trainers' code loop, train/base.py::_code_loop): the lock-step pair path runs once, the following
    encode_image / encode_text calls ON THE SAME TENSORS hand its features out (bit-identical to the separate calls), other tensors
    and later calls take the ordinary path."""
    from model.base.model import CLIP
    cfg = recipe.CLIP_TINY
    torch.manual_seed(3)
    m = CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"], cfg["vision_width"], cfg["vision_patch_size"],
             cfg["context_length"], cfg["vocab_size"], cfg["transformer_width"], cfg["transformer_heads"],
             cfg["transformer_layers"]).to(DEV).float().set_gemm_dtype("f32")
    img = torch.from_numpy(recipe.images(5, cfg["image_resolution"], 1)).to(DEV)
    txt = torch.from_numpy(recipe.captions(5, 16, cfg["vocab_size"], 2)).to(DEV)
    with torch.no_grad():
        ref = (m.encode_image(img).clone(), m.encode_text(txt).clone())
        m.prefetch_pair(img, txt)
        assert set(m._pair_stash) == {"image", "text"}
        other = m.encode_image(img.clone())                        # another tensor object: the ordinary path, stash untouched
        assert torch.equal(other, ref[0]) and "image" in m._pair_stash
        a, b = m.encode_image(img), m.encode_text(txt)
        assert torch.equal(a, ref[0]) and torch.equal(b, ref[1]) and not m._pair_stash
        assert torch.equal(m.encode_image(img), ref[0])            # consumed: the ordinary path again


@pytest.mark.parametrize("cfg_over", [
    dict(vision_width=256, transformer_width=128, transformer_heads=2),      # fp16 stream on the image side only (width % 256)
    dict(vision_width=128, transformer_width=256, transformer_heads=4),      # ... on the text side only
    dict(vision_width=256, transformer_width=256, transformer_heads=4, vision_layers=3, transformer_layers=2),   # depths differ
    dict(vision_width=256, transformer_width=256, transformer_heads=4, vision_layers=1, transformer_layers=3),
])
def test_encode_pair_with_towers_of_different_stream_kinds_and_depths(cfg_over):
    """The lock-step path shares ONE set of epilogue flags per grouped launch: towers whose residual streams differ in kind (fp16 for
    widths that are multiples of 256 in the bf16 mode, f32 otherwise) or whose depths differ must fall back, block by block, to the
    single-tower calls - same bits as encode_image + encode_text (round-4 advisor finding: the pair path took tower a's flags for both)."""
    from model.base.model import CLIP
    cfg = dict(recipe.CLIP_TINY, **cfg_over)
    torch.manual_seed(31)
    m = CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"], cfg["vision_width"], cfg["vision_patch_size"],
             cfg["context_length"], cfg["vocab_size"], cfg["transformer_width"], cfg["transformer_heads"],
             cfg["transformer_layers"]).to(DEV).float().set_gemm_dtype("bf16")
    m.assume_frozen = True
    img = torch.from_numpy(recipe.images(40, cfg["image_resolution"], 1)).to(DEV)
    txt = torch.from_numpy(recipe.captions(40, 16, cfg["vocab_size"], 2)).to(DEV)
    with torch.no_grad():
        for pack in (True, False):
            m.pack_text = pack
            ref = (m.encode_image(img).clone(), m.encode_text(txt).clone())
            got = tuple(t.clone() for t in m.encode_pair(img, txt))
            assert torch.isfinite(ref[0]).all() and torch.isfinite(ref[1]).all()
            assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])


def test_unclaimed_prefetch_is_dropped_and_counted():
    """A work() that takes one side only must not pin the batch: train/base.py::_pipelined_batches drops the stash after every batch
    (CLIP.drop_pair_stash counts what nobody picked up); features computed under other weights are never handed out."""
    from model.base.model import CLIP
    cfg = recipe.CLIP_TINY
    torch.manual_seed(3)
    m = CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"], cfg["vision_width"], cfg["vision_patch_size"],
             cfg["context_length"], cfg["vocab_size"], cfg["transformer_width"], cfg["transformer_heads"],
             cfg["transformer_layers"]).to(DEV).float().set_gemm_dtype("f32")
    img = torch.from_numpy(recipe.images(5, cfg["image_resolution"], 1)).to(DEV)
    txt = torch.from_numpy(recipe.captions(5, 16, cfg["vocab_size"], 2)).to(DEV)
    with torch.no_grad():
        m.prefetch_pair(img, txt)
        _ = m.encode_image(img)
        m.drop_pair_stash()
        assert not m._pair_stash and m.pair_stash_misses == 1
        m.prefetch_pair(img, txt)
        m.visual.proj.mul_(2.0)                                  # the weights change between the prefetch and the use
        fresh = m.encode_image(img)
        assert "image" in m._pair_stash                          # refused: computed under the old weights
        m.drop_pair_stash()
        assert torch.equal(fresh, m.encode_image(img))


def test_alternating_streams_code_loop_equals_the_plain_loop(tmp_path, monkeypatch):
    """train/base.py::_code_loop on two alternating streams with the lock-step pair path == the same loop with CMH_OVERLAP=0
    and CMH_PAIR=0 (one stream, separate encodes): identical code buffers."""
    import argparse
    import os
    import sys
    import dataset.synthetic as ds
    import main
    ck = os.path.join(tmp_path, "clip.pt")
    torch.save({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(dict(recipe.CLIP_TINY, embed_dim=512), 7).items()}, ck)
    monkeypatch.setattr(ds, "SOT", 510)
    monkeypatch.setattr(ds, "EOT", 511)
    monkeypatch.setattr(sys, "argv", ["main.py", "-clip-path", ck, "--save-dir", os.path.join(tmp_path, "run"), "--batch-size", "8",
                                      "--num-workers", "0", "--resolution", "64", "--max-words", "16", "--query-num", "20",
                                      "--train-num", "30", "--synthetic-size", "70", "--gemm-dtype", "f32", "--epochs", "0"])
    torch.manual_seed(1)
    tr = main.trainers["DSPH"](argparse.Namespace(method="DSPH", dataset="synthetic", output_dim=16, is_train=True), 0)
    tr.change_state(mode="valid")
    piped = [t.clone() for t in tr.get_code(tr.retrieval_loader, tr.args.retrieval_num)[:2]]
    monkeypatch.setenv("CMH_OVERLAP", "0")
    monkeypatch.setenv("CMH_PAIR", "0")
    plain = [t.clone() for t in tr.get_code(tr.retrieval_loader, tr.args.retrieval_num)[:2]]
    assert torch.equal(piped[0], plain[0]) and torch.equal(piped[1], plain[1])
    assert set(piped[0].unique().tolist()) <= {-1.0, 0.0, 1.0}
