"""Seeded raw images for the input-pipeline tests (shared by tests/golden/make_golden8.py and the parity tests)."""
import numpy as np

# (H, W, R): up- and down-scaling, both orientations, square, already-R edges, odd crop margins (half-to-even rounding)
CASES = [(37, 53, 16), (53, 37, 16), (100, 75, 32), (20, 31, 32), (15, 15, 32), (32, 45, 32), (33, 32, 32), (61, 32, 32),
         (224, 300, 224), (333, 500, 224), (500, 375, 224), (640, 427, 224)]
SMALL = [c for c in CASES if c[2] < 224]


def image(h, w, seed=0):
    """uint8 RGB [h, w, 3]: smooth gradients + texture + saturated patches (exercises the clip8 at both ends)."""
    rng = np.random.default_rng([seed, h, w])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 11.0), 127 + 120 * np.cos(xx / 5.0 - yy / 13.0), (xx * 255 / max(w - 1, 1))], -1)
    img = base + rng.normal(0, 25, (h, w, 3))
    img[: h // 4, : w // 4] = 255
    img[h - h // 5:, w - w // 5:] = 0
    img[h // 2, :] = rng.integers(0, 2, (w, 3)) * 255
    return np.clip(img, 0, 255).astype(np.uint8)
