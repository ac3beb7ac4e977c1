"""DSPH trainer (reference train/DSPH/hash_train.py:16-73; paper: Deep Semantic-aware Proxy Hashing,
TCSVT 2023).  Forward, loss and validation run on libcmh, and so does the optimiser (fused BertAdam, SURVEY §8f "next"
#1) and the backward through heads and both towers (#2): train_epoch is a complete training loop."""
import os
import time

import torch

from model.DSPH import MDSPH
from model.base.optimization import BertAdam
from train.base import TrainBase
from .get_args import get_args
from .loss import HyP


class DSPHTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(DSPHTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        self.model = MDSPH(outputDim=self.args.output_dim, clipPath=self.args.clip_path,
                           writer=self.writer, logger=self.logger, is_train=self.args.is_train).to(self.rank)
        if self.args.pretrained != "" and os.path.exists(self.args.pretrained):
            self.logger.info("load pretrained model.")
            self.model.load_state_dict(torch.load(self.args.pretrained, map_location=f"cuda:{self.rank}"))
        self.model.float()
        self.model.clip.set_gemm_dtype(self.args.gemm_dtype)
        self.args.numclass = self.args.nclass
        self.hyp = HyP(self.args).to(self.rank)
        # reference train/DSPH/hash_train.py:35-44: two learning rates, warm-up cosine, SGD for the proxies
        self.optimizer = BertAdam([
            {"params": self.model.clip.parameters(), "lr": self.args.clip_lr},
            {"params": self.model.image_hash.parameters(), "lr": self.args.lr},
            {"params": self.model.text_hash.parameters(), "lr": self.args.lr}],
            lr=self.args.lr, warmup=self.args.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
            t_total=len(self.train_loader) * self.args.epochs, weight_decay=self.args.weight_decay, max_grad_norm=1.0)
        self.optimizer_loss = torch.optim.SGD(params=self.hyp.parameters(), lr=0.02, momentum=0.9, weight_decay=0.0005)
        self.total_time = 0

    def compute_loss(self, hash_img, hash_text, label):
        return self.hyp(hash_img, hash_text, label)

    def _step(self, image, text, label):
        """One optimisation step (reference :52-70): forward, HyP loss, backward, BertAdam on the model + SGD on the proxies."""
        dev = self.rank
        image, text, label = (t.to(dev, non_blocking=True) for t in (image, text, label))
        hash_img, hash_text = self.model(image, text)
        # several ranks: ONE fused all-gather of [B_local, 2K + C]; HyP's pairwise terms (loss.py:42-66) see the global batch
        hash_img, hash_text, label = self.loss_inputs(hash_img, hash_text, label.float())
        loss = self.compute_loss(hash_img, hash_text, label)
        for opt in (self.optimizer, self.optimizer_loss):
            opt.zero_grad()
        self.backward(loss, self.hyp)             # + the gradient means over the ranks when there are several
        for opt in (self.optimizer, self.optimizer_loss):
            opt.step()
        return loss

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        all_loss = 0
        for image, text, label, index in self.train_loader:
            began = time.time()
            self.global_step += 1
            all_loss += self._step(image, text, label)
            self.total_time += time.time() - began
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}, time: {self.total_time}")
