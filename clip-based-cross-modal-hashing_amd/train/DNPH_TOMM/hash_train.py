"""DNPH (TOMM) trainer (reference train/DNPH_TOMM/hash_train.py:16-89): forward, loss, backward (heads, classifier and both
towers) and the fused BertAdam step all run on libcmh.  The Hungarian noise assignment stays numpy on the host, like upstream."""
import os
import time

import torch

from model.DNPH_TOMM import MDNPH
from model.base.optimization import BertAdam
from train.base import TrainBase
from .b_reg import gene_noise, rand_unit_rect
from .get_args import get_args
from .loss import DNPH_out


class DNPHTOMMTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(DNPHTOMMTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        self.model = MDNPH(outputDim=self.args.output_dim, num_classes=self.args.nclass, clipPath=self.args.clip_path,
                           writer=self.writer, logger=self.logger, is_train=self.args.is_train).to(self.rank)
        if self.args.pretrained != "" and os.path.exists(self.args.pretrained):
            self.model.load_state_dict(torch.load(self.args.pretrained, map_location=f"cuda:{self.rank}"))
        self.model.float()
        self.model.clip.set_gemm_dtype(self.args.gemm_dtype)
        self.optimizer = BertAdam([
            {"params": self.model.clip.parameters(), "lr": self.args.clip_lr},
            {"params": self.model.image_hash.parameters(), "lr": self.args.lr},
            {"params": self.model.text_hash.parameters(), "lr": self.args.lr},
            {"params": self.model.image_pre.parameters(), "lr": self.args.lr},
            {"params": self.model.text_pre.parameters(), "lr": self.args.lr}],
            lr=self.args.lr, warmup=self.args.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
            t_total=len(self.train_loader) * self.args.epochs, weight_decay=self.args.weight_decay, max_grad_norm=1.0)
        self.DNPH = DNPH_out(self.args).to(self.rank)
        self.total_time = 0
        # upstream builds this SGD for the proxies but never steps it (hash_train.py:48, :83-85): the proxies stay at their seed
        self.optimizer_loss = torch.optim.SGD(params=self.DNPH.parameters(), lr=1e-4)

    def noise_rows(self, hash_img, hash_text):
        """The uniform-distribution regulariser's targets (b_reg.py): one random +-1 matrix per step, assigned to the samples of
        each modality by the Hungarian method on the host (numpy / scipy, as upstream)."""
        s_vector = rand_unit_rect(*hash_img.shape)
        on_host = lambda h: h.cpu().detach().numpy()
        to_dev = lambda a: torch.from_numpy(a).float().to(self.rank)
        return to_dev(gene_noise(on_host(hash_img), s_vector)), to_dev(gene_noise(on_host(hash_text), s_vector))

    def compute_loss(self, hash_img, pre_img, hash_text, pre_text, label):
        i_noises, t_noises = self.noise_rows(hash_img, hash_text)        # Hungarian assignment: host work on the rank's own batch
        # several ranks: ONE fused all-gather of [B_local, 4K + 3C] (hashes, classifier outputs, labels, assigned noise rows);
        # DNPH_out and the noise term are then evaluated on the global batch
        hash_img, pre_img, hash_text, pre_text, label, i_noises, t_noises = self.loss_inputs(
            hash_img, pre_img, hash_text, pre_text, label, i_noises, t_noises)
        return self.DNPH(hash_img, hash_text, pre_img, pre_text, label, label, i_noises, t_noises)

    def _step(self, image, text, label):
        image, text = image.to(self.rank, non_blocking=True), text.to(self.rank, non_blocking=True)
        label = label.to(self.rank, non_blocking=True).float()
        loss = self.compute_loss(*self.model(image, text), label)
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
            all_loss += self._step(image, text, label)
            self.total_time += time.time() - began
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}, time: {self.total_time}")
