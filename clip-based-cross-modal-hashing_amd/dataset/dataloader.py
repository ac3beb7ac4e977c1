"""The reference's on-disk dataset contract (dataset/dataloader.py:6-66): `index.mat["index"]` (image paths, or an .npy of
decoded images with npy=True), `caption.mat["caption"]` or `caption.txt` (one caption per line), `label.mat["category"]`;
seeded permutation -> first query_num = queries, next train_num = training set, everything after the queries = retrieval DB."""
import numpy as np
import scipy.io as scio

from .base import BaseDataset


def split_data(captions, indexs, labels, query_num=5000, train_num=10000, seed=None):
    np.random.seed(seed=seed)
    order = np.random.permutation(range(len(indexs)))
    parts = (order[:query_num], order[query_num:query_num + train_num], order[query_num:])
    return (tuple(indexs[p] for p in parts), tuple(captions[p] for p in parts), tuple(labels[p] for p in parts))


def load_files(captionFile, indexFile, labelFile, npy=False):
    if captionFile.endswith("mat"):
        captions = scio.loadmat(captionFile)["caption"]
        captions = captions[0] if captions.shape[0] == 1 else captions
    elif captionFile.endswith("txt"):
        with open(captionFile, "r") as f:
            captions = np.asarray([[line.strip()] for line in f.readlines()])
    else:
        raise ValueError("the format of 'captionFile' doesn't support, only support [txt, mat] format.")
    indexs = scio.loadmat(indexFile)["index"] if not npy else np.load(indexFile, allow_pickle=True)
    labels = scio.loadmat(labelFile)["category"]
    return captions, indexs, labels


def dataloader(captionFile: str, indexFile: str, labelFile: str, maxWords=32, imageResolution=224, query_num=5000,
               train_num=10000, seed=None, npy=False, dataset_cls=BaseDataset, bpe_path=None):
    captions, indexs, labels = load_files(captionFile, indexFile, labelFile, npy)
    (qi, ti, ri), (qc, tc, rc), (ql, tl, rl) = split_data(captions, indexs, labels, query_num=query_num, train_num=train_num, seed=seed)
    kw = dict(maxWords=maxWords, imageResolution=imageResolution, npy=npy, bpe_path=bpe_path)
    train_data = dataset_cls(captions=tc, indexs=ti, labels=tl, **kw)
    query_data = dataset_cls(captions=qc, indexs=qi, labels=ql, is_train=False, **kw)
    retrieval_data = dataset_cls(captions=rc, indexs=ri, labels=rl, is_train=False, **kw)
    return train_data, query_data, retrieval_data
