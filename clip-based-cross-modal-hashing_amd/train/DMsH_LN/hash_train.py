"""DMsH-LN trainer (reference train/DMsH_LN/hash_train.py:14-76): LinearHash heads on the CLIP towers, a LabelNet that turns the
multi-hot labels into codes, and the multi-similarity loss on image-image, text-text and image-text similarities of the batch -
forward, loss, backward and optimiser on libcmh.  (The reference pins the loss module to `cuda:1`, :35; here everything lives on
the rank's GPU.)"""
import os
import time

import torch

from model.DMsH_LN import MDMsH_LN
from model.base.optimization import BertAdam
from train.base import TrainBase
from .MSLOSS import MultiSimilarityLoss
from .get_args import get_args
from .labelnet import LabelNet
from .loss import HyP  # noqa: F401  (imported by the reference too, :8; never called)


class DMsH_LNTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(DMsH_LNTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        self.model = MDMsH_LN(outputDim=self.args.output_dim, clipPath=self.args.clip_path,
                              writer=self.writer, logger=self.logger, is_train=self.args.is_train).to(self.rank)
        if self.args.pretrained != "" and os.path.exists(self.args.pretrained):
            self.logger.info("load pretrained model.")
            self.model.load_state_dict(torch.load(self.args.pretrained, map_location=f"cuda:{self.rank}"))
        self.model.float()
        self.model.clip.set_gemm_dtype(self.args.gemm_dtype)
        self.MSL = MultiSimilarityLoss().to(self.rank)
        self.L_net = LabelNet(code_len=self.args.output_dim, label_dim=self.args.numclass).to(self.rank)
        self.optimizer = BertAdam([
            {"params": self.model.clip.parameters(), "lr": self.args.clip_lr},
            {"params": self.model.image_hash.parameters(), "lr": self.args.lr},
            {"params": self.model.text_hash.parameters(), "lr": self.args.lr},
            {"params": self.MSL.parameters(), "lr": self.args.lr},
            {"params": self.L_net.parameters(), "lr": self.args.lr}],          # no gradient ever reaches them (labelnet.py)
            lr=self.args.lr, warmup=self.args.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
            t_total=len(self.train_loader) * self.args.epochs, weight_decay=self.args.weight_decay, max_grad_norm=1.0)
        self.total_time = 0

    def compute_loss(self, hash_img, hash_text, label_output):
        """:58-61"""
        img_loss = self.MSL(hash_img, label_output)
        text_loss = self.MSL(hash_text, label_output)
        i_t_loss = self.MSL(hash_img, label_output, feat2=hash_text)
        return img_loss + text_loss + i_t_loss

    def _step(self, image, text, label):
        """One optimisation step (reference :48-70)."""
        image, text = image.to(self.rank, non_blocking=True), text.to(self.rank, non_blocking=True)
        label = label.to(self.rank, non_blocking=True)
        _, _, label_output = self.L_net(label, device=self.rank)
        hash_img, hash_text = self.model(image, text)
        # several ranks: ONE fused all-gather of [B_local, 3K]: the three similarity matrices are B x B in the GLOBAL batch
        hash_img, hash_text, label_output = self.loss_inputs(hash_img, hash_text, label_output)
        loss = self.compute_loss(hash_img, hash_text, label_output)
        self.optimizer.zero_grad()
        self.backward(loss)
        self.optimizer.step()
        return loss

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        all_loss = 0
        self.L_net.set_alpha(epoch)
        for image, text, label, index in self.train_loader:
            began = time.time()
            self.global_step += 1
            all_loss += self._step(image, text, label).detach()
            self.total_time += time.time() - began
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}, lr: "
                         f"{'-'.join([str('%.9f' % itm) for itm in sorted(list(set(self.optimizer.get_lr())))])}, time: {self.total_time}")
