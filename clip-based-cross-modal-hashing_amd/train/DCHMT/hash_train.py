"""DCHMT trainer (reference train/DCHMT/hash_train.py:14-158; paper: Differentiable Cross-modal
Hashing via Multimodal Transformers, ACM MM 2022).  similarity_loss x3 + our_loss are ONE native
call (cmh_dchmt_loss); as for DSPH the backward/optimiser half of the step is not built yet."""
import os

import torch

import cmh_native as N
from model.DCHMT import MDCMHT
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
        self.optimizer = None

    def our_loss(self, image, text, label, epoch=0, times=0):
        """image/text: [B, 2K] pair probabilities (hash_layer == 'select'); label [B,C]."""
        if self.args.hash_layer != "select":
            raise NotImplementedError("hash_layer='linear' calls an undefined self.hash_loss upstream (hash_train.py:131)")
        loss = N.dchmt_loss(image, text, label.to(image.device), self.args.output_dim,
                            self.args.similarity_function, self.args.loss_type, self.args.vartheta,
                            self.args.sim_threshold)
        return no_backward(loss, self.model.image_hash.fc.weight)

    def compute_loss(self, image, text, label, epoch=0, times=0):
        return self.our_loss(image, text, label, epoch, times)

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        all_loss = 0
        times = 0
        for image, text, label, index in self.train_loader:
            self.global_step += 1
            times += 1
            image = image.to(self.rank, non_blocking=True)
            text = text.to(self.rank, non_blocking=True)
            hash_img, hash_text = self.model(image, text)
            hash_img = torch.cat(hash_img, dim=-1) if isinstance(hash_img, list) else hash_img.view(hash_img.shape[0], -1)
            hash_text = torch.cat(hash_text, dim=-1) if isinstance(hash_text, list) else hash_text.view(hash_text.shape[0], -1)
            loss = self.compute_loss(hash_img, hash_text, label, epoch, times)
            all_loss += loss
            loss.backward()      # raises NotImplementedError: backward kernels are the next scope row
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}")
