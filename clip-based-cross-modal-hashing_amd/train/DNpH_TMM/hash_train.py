"""DNpH trainer (reference train/DNpH_TMM/hash_train.py:16-72; paper: Deep Neighborhood-Preserving Hashing With Quadratic Spherical
Mutual Information, TMM 2024): LinearHash heads on the CLIP towers, qmi_loss on the batch, fused BertAdam - forward, loss,
backward and optimiser on libcmh."""
import os
import time

import torch

from model.DNpH_TMM import MDNpH
from model.base.optimization import BertAdam
from train.base import TrainBase
from .get_args import get_args
from .loss import qmi_loss


class DNpHTMMTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(DNpHTMMTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        self.model = MDNpH(outputDim=self.args.output_dim, clipPath=self.args.clip_path,
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
        self.total_time = 0

    def compute_loss(self, hash_img, hash_text, label):
        return qmi_loss(images=hash_img, texts=hash_text, targets=label)

    def _step(self, image, text, label):
        """One optimisation step (reference :49-68)."""
        image, text = image.to(self.rank, non_blocking=True), text.to(self.rank, non_blocking=True)
        label = label.to(self.rank, non_blocking=True).float()
        hash_img, hash_text = self.model(image, text)
        # several ranks: ONE fused all-gather of [B_local, 2K + C]: Y, T, YT and the indicator D are B x B in the GLOBAL batch
        hash_img, hash_text, label = self.loss_inputs(hash_img, hash_text, label)
        loss = self.compute_loss(hash_img, hash_text, label)
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
            self.global_step += 1
            all_loss += self._step(image, text, label).detach()
            self.total_time += time.time() - began
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}, time: {self.total_time}")
