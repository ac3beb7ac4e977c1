"""oracle/preprocess_oracle.py (numpy restatement of torchvision's transform chain over Pillow's resampler) against the
outputs Pillow itself produced (tests/golden/make_golden8.py) — bit for bit, uint8 stage and float tensor."""
import hashlib

import numpy as np
import pytest

import preputil as pu
from oracle import preprocess_oracle as po


@pytest.mark.parametrize("h,w,R", pu.CASES)
@pytest.mark.parametrize("train", [True, False])
def test_transform_matches_pillow(golden, h, w, R, train):
    g = golden("preprocess.npz")
    tag = f"H{h}_W{w}_R{R}_{'train' if train else 'eval'}"
    img = pu.image(h, w)
    u8 = po.transform_u8(img, R, train)
    assert u8.shape == (R, R, 3)
    if f"{tag}_u8" in g.files:
        assert np.array_equal(u8, g[f"{tag}_u8"])
    assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).hexdigest() == str(g[f"{tag}_sha"])
    f = po.transform(img, R, train)
    assert f.dtype == np.float32 and f.shape == (3, R, R)
    if f"{tag}_f32" in g.files:
        assert np.array_equal(f, g[f"{tag}_f32"])
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(g[f"{tag}_fsha"])


def test_crop_origin_rounds_half_to_even():
    assert po.crop_origin(225, 229, 224) == (0, 2)          # 0.5 -> 0, 2.5 -> 2
    assert po.crop_origin(227, 224, 224) == (2, 0)          # 1.5 -> 2
    assert po.resized_size(375, 500, 224, True) == (224, 298) and po.resized_size(500, 375, 224, True) == (298, 224)
