"""BaseDataset with the reference's constructor and item contract (dataset/base.py:14-106), re-cut for a GPU-side input pipeline.

Upstream `__getitem__` does everything per item on the CPU: PIL decode -> bicubic resize / crop / normalise -> float32
[3, R, R], and the Python BPE tokenizer.  Here an item carries the DECODED uint8 image and the chosen caption STRING; the
batch is finished by `collate` (native batch tokenizer -> int64 [B, maxWords], images packed into one ragged uint8 buffer)
and by `DeviceLoader`, which uploads the raw pixels and runs `cmh_image_preprocess` on the GPU.  What the trainer iterates
over is unchanged: (image f32 [B, 3, R, R] on the device, caption int64 [B, maxWords], label, index) — bit-identical images
(tests/test_gpu_preprocess.py) and identical token ids (tests/test_tokenizer.py)."""
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from dataset.gpu_transform import RaggedImages, normalize_u8, preprocess

_TOKENIZER = {}


def shared_tokenizer(bpe_path=None):
    """One SimpleTokenizer per process and merges file (DataLoader workers build their own after the fork)."""
    import os
    key = (os.getpid(), bpe_path)
    if key not in _TOKENIZER:
        from model.base.simple_tokenizer import SimpleTokenizer
        _TOKENIZER[key] = SimpleTokenizer(bpe_path)
    return _TOKENIZER[key]


class BaseDataset(Dataset):

    def __init__(self, captions, indexs, labels, is_train=True, tokenizer=None, maxWords=32, imageResolution=224, npy=False,
                 bpe_path=None):
        self.captions, self.indexs, self.labels, self.npy = captions, indexs, labels, npy
        self.maxWords, self.imageResolution, self.is_train = maxWords, imageResolution, is_train
        self._tokenizer, self.bpe_path = tokenizer, bpe_path
        self.SPECIAL_TOKEN = {"CLS_TOKEN": "<|startoftext|>", "SEP_TOKEN": "<|endoftext|>",
                              "MASK_TOKEN": "[MASK]", "UNK_TOKEN": "[UNK]", "PAD_TOKEN": "[PAD]"}
        self.__length = len(self.indexs)

    @property
    def tokenizer(self):
        return self._tokenizer if self._tokenizer is not None else shared_tokenizer(self.bpe_path)

    def __len__(self):
        return self.__length

    def _load_image(self, index: int) -> torch.Tensor:
        """-> uint8 [H, W, 3]: what `Image.open(path).convert("RGB")` / `Image.fromarray(a).convert("RGB")` hold (base.py:55-62)."""
        if not self.npy:
            from PIL import Image
            arr = np.array(Image.open(str(self.indexs[index]).strip()).convert("RGB"))
        else:
            arr = np.asarray(self.indexs[index])
            if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
                from PIL import Image
                arr = np.asarray(Image.fromarray(arr).convert("RGB"))
        return torch.from_numpy(np.ascontiguousarray(arr))

    @staticmethod
    def _caption_list(entry):
        """The captions of one item as a list of str.  Upstream indexes `captions[index]` directly, which works for string
        matrices ([N, n_captions] '<U', blank padded) and for caption.txt ([N, 1]); MATLAB cell arrays come back from
        scipy.io.loadmat as nested object arrays ([[array(['a dog']), array(['two cats'])]]) — those are flattened here."""
        if isinstance(entry, (str, bytes, np.str_)):
            return [entry.decode() if isinstance(entry, bytes) else str(entry)]
        out = []
        for e in (entry.ravel() if isinstance(entry, np.ndarray) else entry):
            out.extend(BaseDataset._caption_list(e))
        return out

    def _choose_caption(self, index: int) -> str:
        captions = self._caption_list(self.captions[index])
        return captions[random.randint(0, len(captions) - 1)]                  # base.py:67-68

    def _load_text(self, index: int):
        """One caption's ids (upstream's per-item path; the loader uses the batch call in `collate`)."""
        return self.tokenizer.encode_captions([self._choose_caption(index)], self.maxWords)[0]

    def _load_label(self, index: int) -> torch.Tensor:
        return torch.from_numpy(np.asarray(self.labels[index]))

    def get_all_label(self):
        labels = torch.zeros([self.__length, len(self.labels[0])], dtype=torch.float32)
        for i, item in enumerate(self.labels):
            labels[i] = torch.from_numpy(np.asarray(item))
        return labels

    def __getitem__(self, index):
        return self._load_image(index), self._choose_caption(index), self._load_label(index), index

    # ---- batch side ----------------------------------------------------------------------------------------------------------
    def collate(self, items):
        images, captions, labels, index = zip(*items)
        return (RaggedImages.from_arrays(images, pin=False), self.tokenizer.encode_captions(list(captions), self.maxWords),
                torch.stack(labels), torch.tensor(index))

    def cached_batch(self, image, caption, label, index):
        """The tuple a batch served from the device-resident image cache leaves as (DeviceLoader)."""
        return image, caption, label, index

    def finish(self, batch, device):
        """collated batch -> the tuple upstream's loader yields, image already preprocessed on `device`."""
        ragged, caption, label, index = batch
        return preprocess(ragged.to(device), self.imageResolution, self.is_train), caption, label, index


class DeviceLoader:
    """DataLoader over a BaseDataset whose batches leave as upstream's (image, caption, label, index) with the image
    transform done on the GPU.  `len`, `dataset`, iteration: like the DataLoader the trainers hold.

    cache_images (default on): both transform chains are deterministic, so the resized uint8 image of every item is kept in HBM
    the first time it is produced (224 x 224 x 3 bytes each: 3.8 GB for a 25 k-image set, 29 GB for NUS-WIDE's 190 k — MI355X has
    288 GB) and once the set is complete an epoch is: shuffle, pick captions, native tokenizer, gather + normalise on the device —
    no decode, no resize, no PCIe traffic for pixels.  Upstream decodes and resizes every image again in each of its 200 epochs.
    Cached batches equal uncached ones bit for bit (tests/test_gpu_input_pipeline.py)."""

    def __init__(self, dataset, device, cache_images=True, **loader_kwargs):
        self.dataset, self.device = dataset, device
        loader_kwargs.pop("pin_memory", None)                  # the ragged buffer is pinned here, after the workers' pickling
        self.batch_size = loader_kwargs.get("batch_size", 1)
        self.loader = DataLoader(dataset=dataset, collate_fn=dataset.collate, **loader_kwargs)
        self.cache_images = cache_images
        self._cache = None                                      # uint8 [N, R, R, 3] on the device
        self._filled = np.zeros(len(dataset), dtype=bool)
        self._labels = None
        self.cached_epochs = 0

    def __len__(self):
        return len(self.loader)

    def _finish_and_fill(self, batch):
        ds = self.dataset
        ragged, index = batch[0], batch[-1]
        if not self.cache_images:
            return ds.finish(batch, self.device)
        R = ds.imageResolution
        if self._cache is None:
            self._cache = torch.empty(len(ds), R, R, 3, dtype=torch.uint8, device=self.device)
        image, u8 = preprocess(ragged.to(self.device), R, ds.is_train, want_u8=True)
        self._cache[index.to(self.device)] = u8
        self._filled[index.numpy()] = True
        # the subclass may carry extra fields (MITH: key_padding_mask); swap a finished image into its tuple
        return (image,) + tuple(batch[1:])

    def _cached_epoch(self, batches):
        ds = self.dataset
        if self._labels is None:
            self._labels = ds.get_all_label()
        for batch_index in batches:
            index = torch.as_tensor(batch_index, dtype=torch.int64)
            caption = ds.tokenizer.encode_captions([ds._choose_caption(int(i)) for i in index], ds.maxWords)
            label = torch.stack([ds._load_label(int(i)) for i in index])
            image = normalize_u8(self._cache, index.to(self.device))
            yield ds.cached_batch(image, caption, label, index)

    def __iter__(self):
        if self.cache_images:
            # this epoch's batches in the DataLoader's own order (shuffle / sampler / drop_last as given); when every image they
            # name is already resident the epoch is served from HBM — under a DistributedSampler a rank's share, not the whole set
            batches = list(self.loader.batch_sampler)
            if batches and self._filled[np.concatenate([np.asarray(b, dtype=np.int64) for b in batches])].all():
                self.cached_epochs += 1
                yield from self._cached_epoch(batches)
                return
        for batch in self.loader:
            ragged = batch[0]
            slot = None
            if torch.cuda.is_available() and not ragged.pixels.is_pinned():
                ragged.pixels, slot = self._stage(ragged.pixels)
            out = self._finish_and_fill(batch)
            if slot is not None:        # the H2D copy of this staging buffer is queued: it may be refilled once that has run
                self._stage_events[slot].record(torch.cuda.current_stream(self.device))
            yield out

    def _stage(self, pixels):
        """The worker's batch (a shared-memory tensor) -> one of two persistent pinned staging buffers.  Round 1 called
        `pixels.pin_memory()` per batch: a fresh 144 MB page-locked allocation (hipHostMalloc: mmap + lock + GPU mapping) and a copy,
        every batch; with forked DataLoader workers alive and several GB mapped in the parent that allocation is what took
        0.2-0.4 s (each one changes the parent's address space while the children share it; with idle workers or a small parent
        it was 3-7 ms).  Two grow-only buffers, alternated, each guarded by an event recorded after its H2D copy was queued."""
        if not hasattr(self, "_stage_bufs"):
            self._stage_bufs, self._stage_events, self._stage_next = [None, None], [None, None], 0
        i = self._stage_next
        self._stage_next = 1 - i
        n = pixels.numel()
        if self._stage_events[i] is not None:
            self._stage_events[i].synchronize()
        else:
            self._stage_events[i] = torch.cuda.Event()
        if self._stage_bufs[i] is None or self._stage_bufs[i].numel() < n:
            self._stage_bufs[i] = torch.empty(int(n * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        view = self._stage_bufs[i][:n]
        view.copy_(pixels)
        return view, i
