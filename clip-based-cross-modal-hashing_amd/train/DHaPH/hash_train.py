"""DHaPH trainer (reference train/DHaPH/hash_train.py:19-92; paper: Deep Hierarchy-aware Proxy Hashing with Self-paced Learning,
TKDE 2024) - the part of it that trains what is evaluated: LinearHash heads on the CLIP towers and the three self-paced contrastive
losses (image-image, text-text, image-text) of a step, fused BertAdam; forward, loss, backward and optimiser on libcmh.

NOT built: the hyperbolic side model (`HPmodel` + `HPLoss` over `pmath.py`, reference :49-52, :72-74).  Upstream feeds it
`hash_img.detach()` / `hash_text.detach()` (:72-73), so `alpha * loss4` sends no gradient to CLIP or the hash heads and only trains
its own proxies with its own AdamW: the hash codes, the checkpoints `model-<epoch>.pth` and every mAP of a run are the same with and
without it.  What differs is the logged number: the epoch loss here is loss1 + loss2 + loss3."""
import os
import time

import torch

from model.DHaPH import MDHaPH
from model.base.optimization import BertAdam
from train.base import TrainBase
from .MSLoss import MSLoss
from .get_args import get_args


class DHaPHTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(DHaPHTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        self.model = MDHaPH(outputDim=self.args.output_dim, clipPath=self.args.clip_path,
                            writer=self.writer, logger=self.logger, is_train=self.args.is_train).to(self.rank)
        if self.args.pretrained != "" and os.path.exists(self.args.pretrained):
            self.logger.info("load pretrained model.")
            self.model.load_state_dict(torch.load(self.args.pretrained, map_location=f"cuda:{self.rank}"))
        self.model.float()
        self.model.clip.set_gemm_dtype(self.args.gemm_dtype)
        self.optimizer = BertAdam([
            {"params": self.model.clip.parameters(), "lr": self.args.clip_lr},
            {"params": self.model.image_hash.parameters(), "lr": self.args.lr},
            {"params": self.model.text_hash.parameters(), "lr": self.args.lr}],
            lr=self.args.lr, warmup=self.args.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
            t_total=len(self.train_loader) * self.args.epochs, weight_decay=self.args.weight_decay, max_grad_norm=1.0)
        self.msloss = MSLoss(temperature=self.args.tau, totalepoch=self.args.epochs, self_paced=True)
        self.logger.info("DHaPH: the hyperbolic proxy branch (HPmodel / HPLoss) is not built; it is detached from the hash model upstream")
        self.total_time = 0.0

    def compute_loss(self, hash_img, hash_text, label, epoch):
        """:68-70 (the trainer passes epoch + 1)"""
        return (self.msloss(hash_img, hash_img, label, epoch) + self.msloss(hash_text, hash_text, label, epoch) +
                self.msloss(hash_img, hash_text, label, epoch))

    def _step(self, image, text, label, epoch=0):
        """One optimisation step (reference :57-87)."""
        image, text = image.to(self.rank, non_blocking=True), text.to(self.rank, non_blocking=True)
        label = label.to(self.rank, non_blocking=True).float()
        hash_img, hash_text = self.model(image, text)
        # several ranks: ONE fused all-gather of [B_local, 2K + C]: the three cosine matrices are B x B in the GLOBAL batch
        hash_img, hash_text, label = self.loss_inputs(hash_img, hash_text, label)
        loss = self.compute_loss(hash_img, hash_text, label, epoch + 1)
        self.optimizer.zero_grad()
        self.backward(loss)
        self.optimizer.step()
        return loss

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        all_loss = 0
        for image, text, label, index in self.train_loader:
            began = time.time()
            all_loss += self._step(image, text, label, epoch).detach()
            self.total_time += time.time() - began
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}, time: {self.total_time}")
