"""DCHMT trainer (reference train/DCHMT/hash_train.py:14-158; paper: Differentiable Cross-modal
Hashing via Multimodal Transformers, ACM MM 2022).  similarity_loss x3 + our_loss are ONE native
call (cmh_dchmt_loss); its backward, the select head's and both towers' are native too, the optimiser is the fused
BertAdam: train_epoch is the reference's loop (hash_train.py:44-68)."""
import os

import torch

import cmh_native as N
from model.DCHMT import MDCMHT
from model.base.optimization import BertAdam
from model.base.model import no_backward
from train.base import TrainBase
from .get_args import get_args


class DCHMTTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(DCHMTTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        self.model = MDCMHT(outputDim=self.args.output_dim, clipPath=self.args.clip_path,
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

    def our_loss(self, image, text, label, epoch=0, times=0):
        """image/text: [B, 2K] pair probabilities (hash_layer == 'select'); label [B,C]."""
        if self.args.hash_layer != "select":
            raise NotImplementedError("hash_layer='linear' calls an undefined self.hash_loss upstream (hash_train.py:131)")
        label = label.to(image.device)
        if torch.is_grad_enabled() and (image.requires_grad or text.requires_grad):
            from backward_ops import DchmtLoss
            return DchmtLoss.apply(image, text, label, self.args.output_dim, self.args.similarity_function, self.args.loss_type,
                                   self.args.vartheta, self.args.sim_threshold)
        return N.dchmt_loss(image, text, label, self.args.output_dim, self.args.similarity_function, self.args.loss_type,
                            self.args.vartheta, self.args.sim_threshold)

    def compute_loss(self, image, text, label, epoch=0, times=0):
        return self.our_loss(image, text, label, epoch, times)

    def _step(self, image, text, label, epoch=0, times=0):
        """One optimisation step (reference hash_train.py:52-66)."""
        image = image.to(self.rank, non_blocking=True)
        text = text.to(self.rank, non_blocking=True)
        hash_img, hash_text = self.model(image, text)
        hash_img = torch.cat(hash_img, dim=-1) if isinstance(hash_img, list) else hash_img.view(hash_img.shape[0], -1)
        hash_text = torch.cat(hash_text, dim=-1) if isinstance(hash_text, list) else hash_text.view(hash_text.shape[0], -1)
        # several ranks: ONE fused all-gather of [B_local, 4K + C]; similarity_loss (reference :82-114) is O(B^2) in the global batch
        hash_img, hash_text, label = self.loss_inputs(hash_img, hash_text, label.to(self.rank).float())
        loss = self.compute_loss(hash_img, hash_text, label, epoch, times)
        self.optimizer.zero_grad()
        self.backward(loss)                   # + the gradient means over the ranks when there are several
        self.optimizer.step()
        return loss

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        all_loss = 0
        times = 0
        for image, text, label, index in self.train_loader:
            self.global_step += 1
            times += 1
            all_loss += self._step(image, text, label, epoch, times).detach()
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}")
