"""DNPH (TOMM) model (reference model/DNPH_TOMM.py:7-51): LinearHash + a classifier on the same features."""
import logging

import torch
import torch.nn as nn

import cmh_native as N
from model.modelbase import Baseclip
from streams import overlapped


class Pre_Layer(nn.Module):
    def __init__(self, inputdim=2048, nb_class=64):
        super(Pre_Layer, self).__init__()
        self.fc = nn.Linear(inputdim, nb_class)

    def forward(self, data):
        if torch.is_grad_enabled() and (data.requires_grad or self.fc.weight.requires_grad):
            from backward_ops import LinearAct
            return LinearAct.apply(data, self.fc.weight, self.fc.bias, N.ACT_NONE, None, 0.0)
        return N.linear_act(data, self.fc.weight, self.fc.bias, N.ACT_NONE)


class MDNPH(Baseclip):

    def __init__(self, outputDim=64, num_classes=80, clipPath="./ViT-B-32.pt", writer=None,
                 saveDir="./result/log", logger: logging.Logger = None, is_train=True):
        super(MDNPH, self).__init__(outputDim=outputDim, clipPath=clipPath, writer=writer,
                                    saveDir=saveDir, logger=logger, is_train=is_train)
        self.image_pre = Pre_Layer(inputdim=self.embedDim, nb_class=num_classes)
        self.text_pre = Pre_Layer(inputdim=self.embedDim, nb_class=num_classes)

    def encode_image(self, image):
        image_fea = self.clip.encode_image(image)
        return self.image_hash(image_fea), self.image_pre(image_fea)

    def encode_text(self, text):
        text_fea = self.clip.encode_text(text)
        return self.text_hash(text_fea), self.text_pre(text_fea)

    def forward(self, image, text):
        (image_embed, image_pre), (text_embed, text_pre) = overlapped(
            lambda: self.encode_image(image), lambda: self.encode_text(text))
        return image_embed, image_pre, text_embed, text_pre
