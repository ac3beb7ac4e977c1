"""The reference's image transform chains (dataset/base.py:35-44) for a whole batch on the GPU.

Upstream every DataLoader worker runs Resize(BICUBIC) / CenterCrop / ToTensor / Normalize per image on the CPU and ships
float32 [3, R, R] tensors (602 KB per image at R = 224); eight such workers feed a few hundred images per second.  Here the
workers only decode (uint8 HWC, what `Image.open(...).convert("RGB")` holds), the batch travels as one ragged uint8 buffer and
`cmh_image_preprocess` produces the float batch on the device — bit-identical to the upstream chain (Pillow's resampler and
torchvision's size / crop rules, tests/golden/preprocess.npz)."""
import ctypes as C

import numpy as np
import torch

import cmh_native as N

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


class RaggedImages:
    """A batch of decoded RGB images of different sizes: `pixels` uint8 [sum H*W*3] (pinned when built on the host),
    `offsets` int64 [B], `hw` int32 [B, 2]."""

    def __init__(self, pixels, offsets, hw, max_h, max_w):
        self.pixels, self.offsets, self.hw, self.max_h, self.max_w = pixels, offsets, hw, int(max_h), int(max_w)

    def __len__(self):
        return self.hw.shape[0]

    @staticmethod
    def from_arrays(arrays, pin=True):
        """arrays: iterable of uint8 [H, W, 3] numpy arrays / tensors."""
        arrays = [a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a) for a in arrays]
        for a in arrays:
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError(f"decoded RGB uint8 [H, W, 3] expected, got {a.dtype} {a.shape}")
        sizes = np.array([a.size for a in arrays], dtype=np.int64)
        offsets = np.concatenate(([0], np.cumsum(sizes)[:-1])).astype(np.int64)
        pixels = torch.empty(int(sizes.sum()), dtype=torch.uint8)
        if pin and torch.cuda.is_available():
            pixels = pixels.pin_memory()
        flat = pixels.numpy()
        for a, o in zip(arrays, offsets):
            flat[o:o + a.size] = a.reshape(-1)
        hw = torch.tensor([[a.shape[0], a.shape[1]] for a in arrays], dtype=torch.int32)
        return RaggedImages(pixels, torch.from_numpy(offsets), hw, int(hw[:, 0].max()), int(hw[:, 1].max()))

    def to(self, device, non_blocking=True):
        return RaggedImages(self.pixels.to(device, non_blocking=non_blocking), self.offsets.to(device, non_blocking=non_blocking),
                            self.hw.to(device, non_blocking=non_blocking), self.max_h, self.max_w)


def preprocess(batch: RaggedImages, resolution=224, train=True, mean=MEAN, std=STD, want_u8=False):
    """-> float32 [B, 3, R, R] on the batch's device (and, with want_u8, the uint8 [B, R, R, 3] images that reach ToTensor)."""
    N.require_gpu(batch.pixels, batch.offsets, batch.hw)
    B, R = len(batch), int(resolution)
    dev = batch.pixels.device
    out = torch.empty(B, 3, R, R, dtype=torch.float32, device=dev)
    u8 = torch.empty(B, R, R, 3, dtype=torch.uint8, device=dev) if want_u8 else None
    ws = N.workspace(N.lib().cmh_image_preprocess_workspace_bytes(B, batch.max_h, batch.max_w, R), dev, "prep")
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    N.check(N.lib().cmh_image_preprocess(N.ptr(batch.pixels), N.ptr(batch.offsets), N.ptr(batch.hw), B, batch.max_h, batch.max_w, R,
                                         1 if train else 0, m, s, N.ptr(out), N.ptr(u8), N.ptr(ws), ws.numel(), N.stream_ptr(dev)),
            "cmh_image_preprocess")
    return (out, u8) if want_u8 else out


def normalize_u8(u8, rows=None, mean=MEAN, std=STD):
    """ToTensor + Normalize of resized uint8 images on the device: u8 [N, R, R, 3]; rows (int64 device tensor) gathers a batch."""
    N.require_gpu(u8, rows)
    if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[1] != u8.shape[2] or u8.shape[3] != 3 or not u8.is_contiguous():
        raise ValueError("uint8 [N, R, R, 3] contiguous expected")
    B, R = (u8.shape[0] if rows is None else rows.numel()), u8.shape[1]
    if rows is not None:
        rows = rows.to(torch.int64).contiguous()
    out = torch.empty(B, 3, R, R, dtype=torch.float32, device=u8.device)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    N.check(N.lib().cmh_image_normalize(N.ptr(u8), N.ptr(rows), B, R, m, s, N.ptr(out), N.stream_ptr(u8.device)), "cmh_image_normalize")
    return out


class GpuTransform:
    """Drop-in for the Compose([...]) of dataset/base.py:35-44, applied per batch: `GpuTransform(224, is_train)(list_of_arrays, device)`."""

    def __init__(self, resolution=224, is_train=True):
        self.resolution, self.is_train = resolution, is_train

    def __call__(self, images, device="cuda:0"):
        batch = images if isinstance(images, RaggedImages) else RaggedImages.from_arrays(images)
        return preprocess(batch.to(device), self.resolution, self.is_train)
