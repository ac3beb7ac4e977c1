"""DCHMT model (reference model/DCHMT.py:8-45).

HashLayer keeps the reference's parameters (`fc` 512->128 and `hash_list` = K two-way Linears, so
checkpoints load strict) but runs as two launches + one pair-softmax: the K Linear(128,2) are one
[2K,128] GEMM.  forward returns the reference's list of K [B,2] tensors (views of one buffer)."""
import logging

import torch
import torch.nn as nn

import cmh_native as N
from model.base.model import no_backward
from model.modelbase import Baseclip, weights_init_kaiming
from streams import overlapped


class HashLayer(nn.Module):
    LINEAR_EMBED = 128
    SIGMOID_ALPH = 10

    def __init__(self, inputDim=2048, outputDim=64):
        super(HashLayer, self).__init__()
        self.fc = nn.Linear(inputDim, self.LINEAR_EMBED)
        self.fc.apply(weights_init_kaiming)
        self.hash_list = nn.ModuleList([nn.Linear(self.LINEAR_EMBED, 2) for _ in range(outputDim)])
        for item in self.hash_list:
            item.apply(weights_init_kaiming)

    def pair_probs(self, data):
        """[B, 2K] pair probabilities (the trainer's torch.cat(list, -1), train/DCHMT/hash_train.py:55-57)."""
        w = torch.cat([l.weight for l in self.hash_list], 0)
        b = torch.cat([l.bias for l in self.hash_list], 0)
        if torch.is_grad_enabled() and (data.requires_grad or self.fc.weight.requires_grad):
            from backward_ops import LinearAct, PairSoftmax      # training: the same three launches, each with its backward
            embed = LinearAct.apply(data, self.fc.weight, self.fc.bias, N.ACT_RELU, None, 0.0)
            return PairSoftmax.apply(LinearAct.apply(embed, w, b, N.ACT_NONE, None, 0.0))
        embed = N.linear_act(data, self.fc.weight, self.fc.bias, N.ACT_RELU)
        z = N.linear_act(embed, w, b, N.ACT_NONE)
        return N.pair_softmax(z)

    def forward(self, data):
        p = self.pair_probs(data)
        return [p[:, 2 * j:2 * j + 2] for j in range(len(self.hash_list))]


class MDCMHT(Baseclip):

    def __init__(self, outputDim=64, clipPath="./ViT-B-32.pt", writer=None, saveDir="./result/log",
                 logger: logging.Logger = None, is_train=True):
        super(MDCMHT, self).__init__(outputDim=outputDim, clipPath=clipPath, writer=writer,
                                     saveDir=saveDir, logger=logger, is_train=is_train)
        self.image_hash = HashLayer(inputDim=self.embedDim, outputDim=outputDim)
        self.text_hash = HashLayer(inputDim=self.embedDim, outputDim=outputDim)

    def forward(self, image, text):
        return overlapped(lambda: self.encode_image(image), lambda: self.encode_text(text))
