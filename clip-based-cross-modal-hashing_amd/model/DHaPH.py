"""DHaPH model (reference model/DHaPH.py:5-19): Baseclip with LinearHash heads, forward = (image hash, text hash)."""
import logging

from model.modelbase import Baseclip
from streams import overlapped


class MDHaPH(Baseclip):

    def __init__(self, outputDim=64, clipPath="./ViT-B-32.pt", writer=None, saveDir="./result/log",
                 logger: logging.Logger = None, is_train=True):
        super(MDHaPH, self).__init__(outputDim=outputDim, clipPath=clipPath, writer=writer,
                                     saveDir=saveDir, logger=logger, is_train=is_train)

    def forward(self, image, text):
        return overlapped(lambda: self.encode_image(image), lambda: self.encode_text(text))
