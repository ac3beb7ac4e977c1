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

    def _both(self, feature, hash_head, classifier):
        """hash code and class logits from one tower feature (reference :33-51)"""
        return hash_head(feature), classifier(feature)

    def encode_image(self, image):
        return self._both(self.clip.encode_image(image), self.image_hash, self.image_pre)

    def encode_text(self, text):
        return self._both(self.clip.encode_text(text), self.text_hash, self.text_pre)

    def forward(self, image, text):
        (image_embed, image_pre), (text_embed, text_pre) = overlapped(
            lambda: self.encode_image(image), lambda: self.encode_text(text))
        return image_embed, image_pre, text_embed, text_pre
