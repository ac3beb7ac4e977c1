"""TEST INFRASTRUCTURE (the checker, never the product).  numpy restatement of DMsH-LN's label network and multi-similarity loss:
  LabelNet.forward       /root/reference/train/DMsH_LN/labelnet.py:13-18   fc1 -> ReLU -> fc2 -> tanh(alpha * .), alpha = sqrt(epoch + 1) (:21)
  MultiSimilarityLoss    /root/reference/train/DMsH_LN/MSLOSS.py:13-55    (the non-cifar10-1 branch the trainer takes)
Pinned by tests/golden/msl.npz = the reference's own outputs (tests/golden/make_golden16.py)."""
import numpy as np


def label_net(lab, w1, b1, w2, b2, epoch):
    feat = np.maximum(lab.astype(np.float32) @ w1.T + b1, 0.0)
    hid = feat @ w2.T + b2
    return feat, hid, np.tanh(np.float32(np.sqrt(epoch + 1.0)) * hid)


def msl_loss(feats, labels, feat2=None, thresh=0.5, margin=0.1, scale_pos=2.0, scale_neg=40.0, eps=1e-5):
    """:13-55.  sim = the rows of feats.feat2^T L2-normalised (F.normalize, eps 1e-12); similar pairs = labels.labels^T > 0."""
    other = feats if feat2 is None else feat2
    s = feats.astype(np.float64) @ other.astype(np.float64).T
    s = s / np.maximum(np.sqrt((s * s).sum(1, keepdims=True)), 1e-12)
    same = (labels.astype(np.float64) @ labels.astype(np.float64).T) > 0
    B = feats.shape[0]
    total = 0.0
    for i in range(B):
        pos_ = s[i][same[i]]
        pos_ = pos_[pos_ < 1 - eps]
        neg_ = s[i][~same[i]]
        if pos_.size == 0 or neg_.size == 0:
            continue
        neg = neg_[neg_ + margin > pos_.min()]
        pos = pos_[pos_ - margin < neg_.max()]
        if neg.size < 1 or pos.size < 1:
            continue
        total += np.log1p(np.exp(-scale_pos * (pos - thresh)).sum()) / scale_pos + np.log1p(np.exp(scale_neg * (neg - thresh)).sum()) / scale_neg
    return np.float32(total / B)


def spl_loss(image_feature, text_feature, labels, epoch, temperature=0.3, totalepoch=100, self_paced=True):
    """DHaPH's self-paced contrastive loss, /root/reference/train/DHaPH/MSLoss.py:13-33 (pinned by tests/golden/spl.npz,
    tests/golden/make_golden17.py)."""
    f = np.float64
    a, b, lab = image_feature.astype(f), text_feature.astype(f), labels.astype(f)
    mask = (lab @ lab.T > 0).astype(f)                                                   # :15
    an = a / np.maximum(np.sqrt((a * a).sum(1, keepdims=True)), 1e-12)
    bn = b / np.maximum(np.sqrt((b * b).sum(1, keepdims=True)), 1e-12)
    s = an @ bn.T                                                                        # :19
    e = np.exp(s / temperature)
    pos, neg = mask * e, (1 - mask) * e                                                  # :21-23
    if self_paced:                                                                       # :25-31
        third = int(totalepoch / 3)
        delta = epoch / third if epoch <= third else 1
        pos = pos * np.exp(-1 - s) ** (delta / 4)
        neg = neg * np.exp(-1 + s) ** delta
    return np.float32((-np.log(pos.sum(1) / (neg.sum(1) + pos.sum(1)))).mean())          # :32-33
