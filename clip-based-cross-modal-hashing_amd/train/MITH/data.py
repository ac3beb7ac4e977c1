"""MITH's dataset variant (reference train/MITH/data.py:9-75): items additionally carry key_padding_mask = (caption == 0)."""
from dataset.base import BaseDataset
from dataset.dataloader import dataloader


class BasDataset(BaseDataset):

    def collate(self, items):
        ragged, caption, label, index = super().collate(items)
        return ragged, caption, caption == 0, label, index

    def finish(self, batch, device):
        ragged, caption, mask, label, index = batch
        image, caption, label, index = super().finish((ragged, caption, label, index), device)
        return image, caption, mask, label, index


    def cached_batch(self, image, caption, label, index):
        return image, caption, caption == 0, label, index


def generate_dataset(captionFile: str, indexFile: str, labelFile: str, maxWords=32, imageResolution=224, query_num=2000,
                     train_num=10000, seed=None, bpe_path=None):
    return dataloader(captionFile, indexFile, labelFile, maxWords, imageResolution, query_num, train_num, seed,
                      dataset_cls=BasDataset, bpe_path=bpe_path)
