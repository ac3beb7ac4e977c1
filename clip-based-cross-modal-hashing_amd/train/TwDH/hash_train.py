"""TwDH trainer (reference train/TwDH/hash_train.py; paper: Two-Step Discrete Hashing, TOMM 2024): long codes from
the two ModalityHash heads, short codes by long_hash @ trans, targets from per-class hash centres.  Forward, loss,
backward (heads, both towers), the fused BertAdam step and the long+short validation run on libcmh."""
import os

import torch

import cmh_native as N
import dist_utils as du
from model.TwDH import MTwDH
from model.base.optimization import BertAdam
from train.base import TrainBase
from utils.calc_utils import calc_map_k_matrix as calc_map_k
from .get_args import get_args


def synthetic_assets(nclass, long_dim, short_dims=(16, 32, 64), seed=0):
    g = torch.Generator().manual_seed(seed)
    pm1 = lambda *s: torch.where(torch.rand(*s, generator=g) < 0.5, -1.0, 1.0)
    long_center = pm1(nclass, long_dim)
    short = {str(s): pm1(nclass, s) for s in short_dims if s < long_dim}
    trans = {k: torch.randn(2 * long_dim, 2 * int(k), generator=g) * (2 * long_dim) ** -0.5 for k in short}
    return long_center, short, trans


class TwDHTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(TwDHTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        if self.args.synthetic_centers:
            lc, sc, tr = synthetic_assets(self.args.nclass, self.args.output_dim, seed=self.args.seed)
        else:
            lc, sc, tr = self.args.long_center, self.args.short_center, self.args.trans_matrix
        self.model = MTwDH(outputDim=self.args.output_dim, clipPath=self.args.clip_path, writer=self.writer,
                           logger=self.logger, is_train=self.args.is_train, long_center=lc, short_center=sc,
                           trans=tr).to(self.rank)
        if self.args.pretrained != "" and os.path.exists(self.args.pretrained):
            self.model.load_state_dict(torch.load(self.args.pretrained, map_location=f"cuda:{self.rank}"))
        self.model.float()
        self.model.clip.set_gemm_dtype(self.args.gemm_dtype)
        self.optimizer = BertAdam([
            {"params": self.model.clip.parameters(), "lr": self.args.clip_lr},
            {"params": self.model.img_hash.parameters(), "lr": self.args.lr},
            {"params": self.model.txt_hash.parameters(), "lr": self.args.lr}],
            lr=self.args.lr, warmup=self.args.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
            t_total=len(self.train_loader) * self.args.epochs, weight_decay=self.args.weight_decay, max_grad_norm=1.0)
        self.distributed = False
        self.max_short, self.best_epoch_short = {}, {}
        for item in self.model.get_short_dims():
            self.max_short.update({item: {"i2t": 0, "t2i": 0}})
            self.best_epoch_short.update({item: {"i2t": 0, "t2i": 0}})

    # ---- reference helpers (hash_train.py:77-122) ----------------------------------------------------------
    def hash_center_multilables(self, labels, Hash_center, random_center=None):
        """+-1 target codes [B,K]; zeros of the class-centre mean are filled from `random_center` (drawn here like the
        reference's torch.randint_like unless injected for parity tests)."""
        dev = torch.device("cuda", self.rank) if isinstance(self.rank, int) else torch.device(self.rank)
        if random_center is None:
            random_center = torch.randint(0, 2, (Hash_center.shape[1],)).float() * 2 - 1
            if du.active():   # one draw per step for the whole (global) batch, as in a single-GPU run: rank 0's
                random_center = du.broadcast_tensor_(random_center.to(dev), 0)
        return N.twdh_targets(labels.to(dev).float(), Hash_center.to(dev).float(), random_center.to(dev).float())

    def hash_convert(self, hash_label):
        """+-1 codes [B,K] -> one-hot pairs [B,2K] (index 1 for positive bits)."""
        pos = (hash_label > 0).float()
        return torch.stack([1 - pos, pos], dim=-1).reshape(hash_label.shape[0], -1)

    def soft_argmax_hash_loss(self, code):
        return N.twdh_loss(code, code, torch.ones(code.shape[0], code.shape[1] // 2, device=code.device))[1]

    def compute_loss(self, long_img_hash, long_txt_hash, short_img_hash, short_txt_hash, labels, indexs, long_center,
                     short_center, random_centers=None):
        rc = random_centers or {}
        if torch.is_grad_enabled() and (long_img_hash.requires_grad or long_txt_hash.requires_grad):
            from backward_ops import TwdhLoss
            terms = TwdhLoss.apply
        else:
            terms = N.twdh_loss
        target = self.hash_center_multilables(labels, long_center, rc.get("long"))
        nce, quan = terms(long_img_hash, long_txt_hash, target)
        loss = nce + self.args.quan_alpha * quan
        for k, v in short_center.items():
            t_k = self.hash_center_multilables(labels, v, rc.get(k))
            nce_k, quan_k = terms(short_img_hash[k], short_txt_hash[k], t_k)
            loss = loss + self.args.low_rate * nce_k + self.args.low_rate * quan_k
        return loss

    def _step(self, image, text, label, index):
        """One optimisation step (reference hash_train.py:49-75)."""
        image = image.to(self.rank, non_blocking=True)
        text = text.to(self.rank, non_blocking=True)
        il, ish, tl, tsh, lc, sc = self.model(image, text)
        if du.active():
            # ONE fused all-gather of the long and short pair probabilities + labels: the BCE / quantisation means run over
            # the global batch (BatchNorm1d of the image head keeps the rank's batch statistics)
            keys = list(ish.keys())
            parts = self.loss_inputs(il, tl, *[ish[k] for k in keys], *[tsh[k] for k in keys], label.to(self.rank).float())
            il, tl, label = parts[0], parts[1], parts[-1]
            ish = dict(zip(keys, parts[2:2 + len(keys)]))
            tsh = dict(zip(keys, parts[2 + len(keys):2 + 2 * len(keys)]))
        loss = self.compute_loss(il, tl, ish, tsh, label, index.numpy(), lc, sc)
        self.optimizer.zero_grad()
        self.backward(loss)
        self.optimizer.step()
        return loss

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        all_loss = 0
        for image, text, label, index in self.train_loader:
            self.global_step += 1
            all_loss += self._step(image, text, label, index)
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] loss: {all_loss.data / (len(self.train_loader))}")

    def make_hash_code(self, code):
        if isinstance(code, list):
            code = torch.cat(code, dim=-1)
        return N.pair_argmax_codes(code.reshape(code.shape[0], -1))

    def get_code(self, data_loader, length: int):
        short_dims = self.model.get_short_dims()
        mk = lambda d: torch.empty(length, d, dtype=torch.float).to(self.rank)
        long_img, long_txt = mk(self.args.output_dim), mk(self.args.output_dim)
        s_img = {str(d): mk(d) for d in short_dims}
        s_txt = {str(d): mk(d) for d in short_dims}
        seen = []

        def to_device(batch):
            image, text, label, index = batch
            out = (image.to(self.rank, non_blocking=True), text.to(self.rank, non_blocking=True), index.to(self.rank))
            seen.append(out[2])
            return out

        def work(image, text, index):
            li, si = self.model.encode_image(image)
            lt, st = self.model.encode_text(text)
            long_img[index, :] = self.make_hash_code(li)
            long_txt[index, :] = self.make_hash_code(lt)
            for k, v in si.items():
                s_img[k][index, :] = self.make_hash_code(v)
            for k, v in st.items():
                s_txt[k][index, :] = self.make_hash_code(v)

        self._pipelined_batches(data_loader, to_device, work)      # two alternating streams, the towers in lock-step (train/base.py)
        self._gather_code_shards(seen, long_img, long_txt, *s_img.values(), *s_txt.values())
        return long_img, long_txt, s_img, s_txt

    def valid(self, epoch, k=None):
        self.logger.info("Valid.")
        q_li, q_lt, q_si, q_st = self.get_code(self.query_loader, self.args.query_num)
        r_li, r_lt, r_si, r_st = self.get_code(self.retrieval_loader, self.args.retrieval_num)
        out = {"long": self.valid_each(epoch, q_li, q_lt, r_li, r_lt, k)}
        for key in q_si:
            out[key] = self.valid_each(epoch, q_si[key], q_st[key], r_si[key], r_st[key], k, short=key)
        return out

    def valid_each(self, epoch, query_img=None, query_txt=None, retrieval_img=None, retrieval_txt=None, k=None, short=None):
        mAPi2t, mAPt2i, mAPi2i, mAPt2t = self._four_maps(query_img, query_txt, retrieval_img, retrieval_txt, k)   # queries sharded over the ranks
        if short is None:
            if self.max_mapi2t < mAPi2t:
                self.best_epoch_i = epoch
            self.max_mapi2t = max(self.max_mapi2t, mAPi2t)
            if self.max_mapt2i < mAPt2i:
                self.best_epoch_t = epoch
            self.max_mapt2i = max(self.max_mapt2i, mAPt2i)
            tag = "Long"
        else:
            s = int(short)
            self.max_short[s]["i2t"] = max(self.max_short[s]["i2t"], mAPi2t)
            self.max_short[s]["t2i"] = max(self.max_short[s]["t2i"], mAPt2i)
            tag = "Short"
        self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}], {tag}, {query_img.shape[-1]} Bit, MAP(i->t): {mAPi2t}, "
                         f"MAP(t->i): {mAPt2i}, MAP(t->t): {mAPt2t}, MAP(i->i): {mAPi2i}")
        return mAPi2t, mAPt2i, mAPi2i, mAPt2t
