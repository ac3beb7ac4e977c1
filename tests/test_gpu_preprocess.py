"""cmh_image_preprocess (C ABI) vs the oracle and the goldens Pillow produced: the uint8 stage and the float tensor are
bit-identical (integer fixed-point resampler; float32 ToTensor / Normalize are single correctly-rounded operations)."""
import hashlib

import numpy as np
import pytest
import torch

import preputil as pu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("train", [True, False])
def test_ragged_batch_matches_pillow_goldens(golden, train):
    from dataset.gpu_transform import RaggedImages, preprocess
    g = golden("preprocess.npz")
    for R in (16, 32, 224):
        cases = [c for c in pu.CASES if c[2] == R]
        batch = RaggedImages.from_arrays([pu.image(h, w) for h, w, _ in cases]).to(DEV)
        out, u8 = preprocess(batch, R, train, want_u8=True)
        out, u8 = out.cpu().numpy(), u8.cpu().numpy()
        for i, (h, w, _) in enumerate(cases):
            tag = f"H{h}_W{w}_R{R}_{'train' if train else 'eval'}"
            if f"{tag}_u8" in g.files:
                assert np.array_equal(u8[i], g[f"{tag}_u8"]), tag
            assert hashlib.sha256(u8[i].tobytes()).hexdigest() == str(g[f"{tag}_sha"]), tag
            if f"{tag}_f32" in g.files:
                assert np.array_equal(out[i], g[f"{tag}_f32"]), tag
            assert hashlib.sha256(out[i].tobytes()).hexdigest() == str(g[f"{tag}_fsha"]), tag


@pytest.mark.parametrize("train", [True, False])
def test_random_sizes_match_oracle(train):
    """Seeded random sizes around the interesting boundaries (up-scaling, 1-pixel crop margins, extreme aspect ratios)."""
    from dataset.gpu_transform import RaggedImages, preprocess
    from oracle import preprocess_oracle as po
    rng = np.random.default_rng(5)
    R = 24
    sizes = [(int(a), int(b)) for a, b in rng.integers(5, 90, (40, 2))] + [(24, 25), (25, 24), (24, 24), (5, 200), (200, 5), (1, 1)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    out, u8 = preprocess(RaggedImages.from_arrays(imgs).to(DEV), R, train, want_u8=True)
    for i, img in enumerate(imgs):
        assert np.array_equal(u8[i].cpu().numpy(), po.transform_u8(img, R, train)), sizes[i]
        assert np.array_equal(out[i].cpu().numpy(), po.transform(img, R, train)), sizes[i]


def test_full_size_batch_properties():
    """BASELINE-size batch (256 images around 500x375 -> 224): size-independent checks — a constant image stays constant
    (weights sum to 2^22 exactly after normalisation only up to rounding: the value may not move by more than the rounding of
    the fixed-point sum), flipping the input flips the output, and per-image results do not depend on the batch they ride in."""
    from dataset.gpu_transform import RaggedImages, preprocess
    rng = np.random.default_rng(9)
    sizes = [(int(rng.integers(300, 520)), int(rng.integers(300, 520))) for _ in range(256)]
    imgs = [pu.image(h, w, seed=i) for i, (h, w) in enumerate(sizes)]
    imgs[0] = np.full((375, 500, 3), 200, dtype=np.uint8)
    batch = RaggedImages.from_arrays(imgs).to(DEV)
    out, u8 = preprocess(batch, 224, True, want_u8=True)
    assert out.shape == (256, 3, 224, 224) and torch.isfinite(out).all()
    assert int(u8[0].min()) == 200 and int(u8[0].max()) == 200
    flipped = RaggedImages.from_arrays([np.ascontiguousarray(a[:, ::-1]) for a in imgs[:8]]).to(DEV)
    _, u8f = preprocess(flipped, 224, True, want_u8=True)
    # a horizontal flip commutes with the resampler when the crop margin is symmetric (even difference) — check those images
    for i in range(8):
        h, w = sizes[i] if i else (375, 500)
        nw = 224 if w <= h else int(224 * w / h)
        if (nw - 224) % 2 == 0:
            assert torch.equal(u8f[i], u8[i].flip(1)), i
    single = RaggedImages.from_arrays(imgs[5:6]).to(DEV)
    assert torch.equal(preprocess(single, 224, True)[0], out[5])


def test_bad_arguments_fail_loudly():
    import cmh_native as N
    from dataset.gpu_transform import RaggedImages, preprocess
    b = RaggedImages.from_arrays([pu.image(20, 30)]).to(DEV)
    b.max_h = 10 ** 6
    with pytest.raises(N.NativeError):
        preprocess(b, 16, True)
    with pytest.raises(ValueError):
        RaggedImages.from_arrays([np.zeros((4, 4), dtype=np.uint8)])
