"""DNPH (TOMM) trainer (reference train/DNPH_TOMM/hash_train.py:16-89): forward, loss, backward (heads, classifier and both
towers) and the fused BertAdam step all run on libcmh.  The Hungarian noise assignment stays numpy on the host, like upstream."""
import os
import time

import torch

import dist_utils as du
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

    def compute_loss(self, hash_img, pre_img, hash_text, pre_text, label):
        s_vector = rand_unit_rect(*hash_img.shape)
        i_noises = torch.from_numpy(gene_noise(hash_img.cpu().detach().numpy(), s_vector)).float().to(self.rank)
        t_noises = torch.from_numpy(gene_noise(hash_text.cpu().detach().numpy(), s_vector)).float().to(self.rank)
        return self.DNPH(hash_img, hash_text, pre_img, pre_text, label, label, i_noises, t_noises)

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        all_loss = 0
        for image, text, label, index in self.train_loader:
            start_time = time.time()
            self.global_step += 1
            image = image.to(self.rank, non_blocking=True)
            text = text.to(self.rank, non_blocking=True)
            label = label.to(self.rank, non_blocking=True).float()
            hash_img, pre_img, hash_text, pre_text = self.model(image, text)
            loss = self.compute_loss(hash_img, pre_img, hash_text, pre_text, label)
            all_loss += loss
            self.optimizer.zero_grad()
            loss.backward()
            if du.world_size() > 1:
                du.allreduce_mean_([p.grad for p in self.model.parameters() if p.grad is not None])
            self.optimizer.step()
            self.total_time += time.time() - start_time
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}, time: {self.total_time}")
