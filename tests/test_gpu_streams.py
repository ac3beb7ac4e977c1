"""Two-tower overlap (streams.py): running the encoders on one HIP stream each must give bit-identical results to running
them back to back, call after call (shared workspaces, allocator reuse across streams)."""
import numpy as np
import pytest
import torch

import recipe

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_overlapped_towers_equal_sequential(monkeypatch):
    from model.base.model import build_model
    from streams import overlapped
    cfg, seed = recipe.CLIP_TINY, 11
    clip = build_model({k: torch.from_numpy(np.asarray(v)) for k, v in recipe.clip_state_dict(cfg, seed).items()}).to(DEV).float()
    clip.assume_frozen = True
    img = torch.from_numpy(recipe.images(6, cfg["image_resolution"], seed)).to(DEV)
    txt = torch.from_numpy(recipe.captions(6, 16, cfg["vocab_size"], seed)).to(DEV)
    with torch.no_grad():
        monkeypatch.setenv("CMH_OVERLAP", "0")
        ri, rt = overlapped(lambda: clip.encode_image(img), lambda: clip.encode_text(txt))
        monkeypatch.setenv("CMH_OVERLAP", "1")
        for _ in range(5):
            oi, ot = overlapped(lambda: clip.encode_image(img), lambda: clip.encode_text(txt))
            junk = torch.randn(1 << 20, device=DEV)   # main-stream work between calls: allocator reuse across streams
            torch.cuda.synchronize()
            assert torch.equal(oi, ri) and torch.equal(ot, rt)
            del junk


def test_vitb32_overlap_and_fp16_stream_consistent(monkeypatch):
    """Full-size towers, bf16 mode (fp16 residual stream): overlapped == sequential bit for bit."""
    from model.base.model import CLIP
    from streams import overlapped
    torch.manual_seed(5)
    clip = CLIP(**recipe.CLIP_VITB32).to(DEV).float().set_gemm_dtype("bf16")
    clip.assume_frozen = True
    img = torch.randn(8, 3, 224, 224, device=DEV)
    txt = torch.from_numpy(recipe.captions(8, 77, recipe.CLIP_VITB32["vocab_size"], 3)).to(DEV)
    with torch.no_grad():
        monkeypatch.setenv("CMH_OVERLAP", "0")
        ri, rt = overlapped(lambda: clip.encode_image(img), lambda: clip.encode_text(txt))
        monkeypatch.setenv("CMH_OVERLAP", "1")
        oi, ot = overlapped(lambda: clip.encode_image(img), lambda: clip.encode_text(txt))
        torch.cuda.synchronize()
    assert torch.isfinite(ri).all() and torch.isfinite(rt).all()
    assert torch.equal(oi, ri) and torch.equal(ot, rt)


def test_packed_encode_text_never_waits_for_the_device():
    """cmh_text_encode_packed keeps the packed row count on the device (LayerNorm and the GEMM kernels read it themselves): the
    call must return while earlier work of its stream is still running, and the count is fetched only when asked for."""
    from model.base.model import CLIP
    torch.manual_seed(6)
    clip = CLIP(**recipe.CLIP_VITB32).to(DEV).float().set_gemm_dtype("bf16")
    clip.assume_frozen = True
    clip.pack_text = True                                 # (the default; CMH_TEXT_PACK=0 in the environment would switch it off)
    txt = torch.from_numpy(recipe.captions(64, 77, recipe.CLIP_VITB32["vocab_size"], 6)).to(DEV)
    with torch.no_grad():
        ref = clip.encode_text(txt)                       # builds the weight copies and the workspace
        torch.cuda.synchronize()
        a = torch.randn(8192, 8192, device=DEV)
        for _ in range(12):                               # a few hundred milliseconds of queued work ahead of the encode
            a = (a @ a).clamp_(-1, 1)
        out = clip.encode_text(txt)
        assert not torch.cuda.current_stream().query(), "encode_text waited for the stream (host synchronisation in the forward path)"
        torch.cuda.synchronize()
    assert torch.equal(out, ref)
    rows, dense = clip.last_text_rows
    assert dense == 64 * 77 and 64 * 3 <= rows < dense
    clip.pack_text = False
    with torch.no_grad():
        assert torch.equal(clip.encode_text(txt), ref)    # dense == packed, bit for bit
