"""The real-dataset path end to end on the GPU: files on disk (index.mat with image paths, caption.txt, label.mat) ->
dataset/dataloader.py split -> BaseDataset items (decoded uint8 + caption string) -> native batch tokenizer + ragged batch ->
cmh_image_preprocess on the device -> trainer.valid() / train_epoch().  Every batch equals what the reference's per-item
CPU chain yields (the oracle restatement of its transforms; token ids from the Python path)."""
import gzip
import sys

import numpy as np
import pytest
import torch

import bpeutil as bu
import preputil as pu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _write_dataset(root, n, nclass):
    import scipy.io as scio
    from PIL import Image
    rng = np.random.default_rng(1)
    paths, caps = [], []
    words = bu.CORPUS.split()
    for i in range(n):
        h, w = int(rng.integers(40, 120)), int(rng.integers(40, 120))
        p = root / f"im{i:03d}.png"
        Image.fromarray(pu.image(h, w, seed=i)).save(p)
        paths.append(str(p))
        caps.append(" ".join(rng.choice(words, size=int(rng.integers(3, 20)))))
    scio.savemat(root / "index.mat", {"index": np.array(paths)})            # char matrix: shorter paths come back blank-padded
    (root / "caption.txt").write_text("\n".join(caps) + "\n")
    lab = (rng.random((n, nclass)) < 0.2).astype(np.float32)
    lab[np.arange(n), rng.integers(0, nclass, n)] = 1
    scio.savemat(root / "label.mat", {"category": lab})
    return paths, caps, lab


def test_real_dataset_trainer_end_to_end(tmp_path, monkeypatch):
    import argparse
    import recipe
    import main
    import model.base.simple_tokenizer as st
    from oracle import preprocess_oracle as po
    from PIL import Image
    gz = tmp_path / "mini.txt.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(bu.MINI_MERGES, "rb").read())
    monkeypatch.setenv("CMH_BPE_VOCAB", str(gz))
    st.default_bpe.cache_clear()
    data = tmp_path / "data"
    data.mkdir()
    n, C = 72, 21
    paths, caps, lab = _write_dataset(data, n, C)
    ck = tmp_path / "clip.pt"
    sd = recipe.clip_state_dict(dict(recipe.CLIP_TINY, embed_dim=512, vocab_size=1024), 7)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ck)
    monkeypatch.setattr(sys, "argv", ["main.py", "-clip-path", str(ck), "--save-dir", str(tmp_path / "out"), "--batch-size", "16",
                                      "--num-workers", "0", "--resolution", "64", "--max-words", "16", "--query-num", "24",
                                      "--train-num", "32", "--epochs", "0", "--data-dir", str(data), "--seed", "5"])
    args = argparse.Namespace(method="DSPH", dataset="nuswide", output_dim=16, is_train=True)
    tr = main.trainers["DSPH"](args, 0)
    assert len(tr.query_loader.dataset) == 24 and len(tr.train_loader.dataset) == 32 and len(tr.retrieval_loader.dataset) == n - 24
    # the split is the reference's: np.random.seed(seed); permutation
    np.random.seed(5)
    order = np.random.permutation(range(n))
    assert torch.equal(tr.query_labels, torch.from_numpy(lab[order[:24]]))
    tok = st.SimpleTokenizer(str(gz))
    # second pass over each loader: served from the device-resident cache of resized images (no decode / resize), same batches
    for loader, ids, train in ((tr.train_loader, order[24:56], True), (tr.query_loader, order[:24], False)) * 2:
        seen = 0
        for image, caption, label, index in loader:
            assert image.is_cuda and image.dtype == torch.float32 and image.shape[1:] == (3, 64, 64) and caption.shape[1] == 16
            for k in range(len(index)):
                src = int(ids[int(index[k])])
                ref = po.transform(np.asarray(Image.open(paths[src]).convert("RGB")), 64, train)
                assert np.array_equal(image[k].cpu().numpy(), ref)
                assert caption[k].tolist() == tok.caption_ids(caps[src], 16)
                assert np.array_equal(label[k].numpy(), lab[src])
            seen += len(index)
        assert seen == len(ids)
    assert tr.train_loader.cached_epochs == 1 and tr.query_loader.cached_epochs == 1 and tr.retrieval_loader.cached_epochs == 0
    tr.change_state(mode="valid")
    maps = tr.valid(0)
    assert all(0.0 <= float(m) <= 1.0 for m in maps[:4])
    for grp in tr.optimizer.param_groups:
        grp["t_total"] = 4
    before = tr.model.image_hash.fc.weight.detach().clone()
    tr.train_epoch(0)
    assert not torch.equal(before, tr.model.image_hash.fc.weight.detach())
    st.default_bpe.cache_clear()


def test_mith_dataset_carries_the_padding_mask(tmp_path, monkeypatch):
    import model.base.simple_tokenizer as st
    from dataset.base import DeviceLoader
    from train.MITH.data import generate_dataset
    gz = tmp_path / "mini.txt.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(bu.MINI_MERGES, "rb").read())
    data = tmp_path / "data"
    data.mkdir()
    _write_dataset(data, 30, 24)
    trn, q, r = generate_dataset(str(data / "caption.txt"), str(data / "index.mat"), str(data / "label.mat"), maxWords=16,
                                 imageResolution=32, query_num=10, train_num=12, seed=1, bpe_path=str(gz))
    loader = DeviceLoader(trn, DEV, batch_size=6, shuffle=False)
    first = [b for b in loader]
    image, caption, mask, label, index = first[0]
    assert image.shape == (6, 3, 32, 32) and torch.equal(mask, caption == 0) and label.shape == (6, 24)
    again = [b for b in loader]                                 # from the cache: same images, same order (no shuffle)
    assert loader.cached_epochs == 1 and len(again) == len(first) == 2
    for a, b in zip(first, again):
        assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]) and torch.equal(b[2], b[1] == 0)
    # a rank's share under a DistributedSampler is served from the cache once THAT share is resident
    from torch.utils.data.distributed import DistributedSampler
    share = DeviceLoader(trn, DEV, batch_size=4, sampler=DistributedSampler(trn, num_replicas=2, rank=1, shuffle=False))
    one, two = [b for b in share], [b for b in share]
    assert share.cached_epochs == 1 and [b[4].tolist() for b in one] == [b[4].tolist() for b in two] == [[1, 3, 5, 7], [9, 11]]
    assert all(torch.equal(a[0], b[0]) for a, b in zip(one, two)) and torch.equal(one[0][0][0], first[0][0][1])
    plain = DeviceLoader(trn, DEV, cache_images=False, batch_size=6, shuffle=False)
    assert torch.equal(next(iter(plain))[0], first[0][0]) and plain.cached_epochs == 0
    # decoding + native tokenisation inside forked DataLoader workers (they never touch the GPU), batches finished here
    forked = [b for b in DeviceLoader(trn, DEV, cache_images=False, batch_size=6, shuffle=False, num_workers=2)]
    assert len(forked) == 2 and all(torch.equal(a[0], b[0]) and torch.equal(a[4], b[4]) for a, b in zip(first, forked))
