"""DNPH (TOMM) trainer (reference train/DNPH_TOMM/hash_train.py:51-89)."""
import os

import torch

from model.DNPH_TOMM import MDNPH
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
        self.DNPH = DNPH_out(self.args).to(self.rank)
        self.optimizer = None

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
            self.global_step += 1
            image = image.to(self.rank, non_blocking=True)
            text = text.to(self.rank, non_blocking=True)
            label = label.to(self.rank, non_blocking=True).float()
            hash_img, pre_img, hash_text, pre_text = self.model(image, text)
            loss = self.compute_loss(hash_img, pre_img, hash_text, pre_text, label)
            all_loss += loss
            loss.backward()      # raises NotImplementedError: backward kernels are the next scope row
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}")
