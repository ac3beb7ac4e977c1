"""Model API of the reference (model/modelbase.py:10-96) on top of libcmh.so.

  weights_init_kaiming  :10-22
  LinearHash            :25-35   fc -> Dropout(0.2) -> tanh   (one fused launch: cmh_linear_act)
  Baseclip              :38-96   load_clip / encode_image / encode_text / forward / eval / train
Same constructor, attributes (.clip, .image_hash, .text_hash, .embedDim) and state_dict keys
(`clip.*`, `image_hash.fc.{weight,bias}`, `text_hash.fc.{weight,bias}`).
"""
import logging
import os

import torch
import torch.nn as nn

import cmh_native as N
from model.base.model import build_model, no_backward
from utils import get_logger, get_summary_writer
from streams import overlapped


def weights_init_kaiming(m):
    """The reference's initialiser (model/modelbase.py:10-22), dispatched on the module's class name like upstream: Linear ->
    kaiming-uniform fan_out + zero bias, Conv -> kaiming-normal fan_in (+ zero bias), affine BatchNorm -> weight 1 / bias 0."""
    kind = type(m).__name__
    if 'Linear' in kind:
        nn.init.kaiming_uniform_(m.weight, mode='fan_out')
        nn.init.zeros_(m.bias)
    elif 'Conv' in kind:
        nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in')
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif 'BatchNorm' in kind and m.affine:
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


class LinearHash(nn.Module):

    def __init__(self, inputDim=2048, outputDim=64):
        super(LinearHash, self).__init__()
        self.fc = nn.Linear(inputDim, outputDim)
        self.fc.apply(weights_init_kaiming)
        self.drop_out = nn.Dropout(p=0.2)

    def forward(self, data, drop_mask=None):
        """tanh(dropout(fc(data))).  In training mode a Bernoulli keep-mask is drawn on the device
        (or injected through `drop_mask` for parity tests: GPU RNG != CPU RNG)."""
        if self.training and drop_mask is None and self.drop_out.p > 0:
            drop_mask = (torch.rand(data.shape[0], self.fc.out_features, device=data.device)
                         >= self.drop_out.p).float()
        if torch.is_grad_enabled() and (data.requires_grad or self.fc.weight.requires_grad):
            from backward_ops import LinearAct
            return LinearAct.apply(data, self.fc.weight, self.fc.bias, N.ACT_TANH, drop_mask, self.drop_out.p)
        return N.linear_act(data, self.fc.weight, self.fc.bias, N.ACT_TANH, drop_mask, self.drop_out.p)


class Baseclip(nn.Module):
    """CLIP trunk + one LinearHash per modality.  Constructor signature, attribute names and state_dict keys are the reference's."""

    def __init__(self, outputDim=64, clipPath="./ViT-B-32.pt", writer=None, saveDir="./result/log",
                 logger: logging.Logger = None, is_train=True):
        super().__init__()
        os.makedirs(saveDir, exist_ok=True)
        if logger is None:
            logger = get_logger(os.path.join(saveDir, "train.log" if is_train else "test.log"))
        if writer is None or not is_train:
            writer = get_summary_writer(os.path.join(saveDir, "tensorboard"))
        self.logger, self.writer = logger, writer
        self.embedDim, self.clip = self.load_clip(clipPath)
        self.image_hash, self.text_hash = (LinearHash(inputDim=self.embedDim, outputDim=outputDim) for _ in range(2))

    def load_clip(self, clipPath) -> tuple:
        """-> (embed_dim, CLIP) from an OpenAI JIT archive, a plain state_dict file, or a state_dict (random-init runs)."""
        if isinstance(clipPath, dict):
            state_dict = dict(clipPath)
        else:
            try:
                state_dict = torch.jit.load(clipPath, map_location="cpu").eval().state_dict()
            except RuntimeError:                       # not a TorchScript archive
                state_dict = torch.load(clipPath, map_location="cpu")
        return state_dict["text_projection"].shape[1], build_model(state_dict)

    # upstream's eval() / train() toggle the two heads only (the trunk has no mode-dependent layers) and return None
    def eval(self):
        for head in (self.image_hash, self.text_hash):
            head.eval()

    def train(self):
        for head in (self.image_hash, self.text_hash):
            head.train()

    def encode_image(self, image):
        return self.image_hash(self.clip.encode_image(image))

    def encode_text(self, text):
        return self.text_hash(self.clip.encode_text(text))

    def forward(self, image, text):
        # the two towers are independent until the loss: one HIP stream each (streams.py)
        return overlapped(lambda: self.encode_image(image), lambda: self.encode_text(text))
