"""MITH trainer (reference train/MITH/hash_train.py; paper: Multi-Granularity Interactive Transformer Hashing, ACM MM 2023).
Forward (token-returning trunk + HashingModel), the memory-bank update, all five loss groups, their backward through the
HashingModel and both towers, the fused BertAdam step and validation run on libcmh.  Under data parallelism the bank update needs every rank's (index, codes):
dist_utils.all_gather_rows + scatter_by_index."""
import os

import torch

import cmh_native as N
import dist_utils as du
import mith_ops as M
from model.MITH import MITH
from model.base.optimization import BertAdam
from train.base import TrainBase
from .get_args import get_args


class MITHTrainer(TrainBase):

    def __init__(self, args, rank=0):
        args = get_args(args)
        args.rank = rank
        super(MITHTrainer, self).__init__(args)
        self.logger.info("dataset len: {}".format(len(self.train_loader.dataset)))
        self.run()

    def _init_model(self):
        self.logger.info("init model.")
        self.model = MITH(args=self.args).to(self.rank)
        if self.args.pretrained != "" and os.path.exists(self.args.pretrained):
            self.model.load_state_dict(torch.load(self.args.pretrained, map_location=f"cuda:{self.rank}"))
        self.model.float()
        self.model.clip.set_gemm_dtype(self.args.gemm_dtype)
        self.model.hash.set_gemm_dtype(self.args.gemm_dtype)
        # reference train/MITH/hash_train.py:36-42: the trunk at clip_lr, the HashingModel at lr
        self.optimizer = BertAdam([
            {"params": self.model.clip.parameters(), "lr": self.args.clip_lr},
            {"params": self.model.hash.parameters(), "lr": self.args.lr}],
            lr=self.args.lr, warmup=self.args.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
            t_total=len(self.train_loader) * self.args.epochs, weight_decay=self.args.weight_decay, max_grad_norm=1.0)
        self.k_bits = self.args.output_dim
        n = len(self.train_loader.dataset)
        mk = lambda: torch.randn(n, self.k_bits).to(self.rank, non_blocking=True)
        self.img_buffer_tokens, self.img_buffer_cls = mk(), mk()
        self.txt_buffer_tokens, self.txt_buffer_cls = mk(), mk()

    def change_state(self, mode):
        self.model.train() if mode == "train" else self.model.eval()

    # ---- reference loss helpers (:103-147), each ONE native call ---------------------------------------
    @staticmethod
    def _grad(*ts):
        return torch.is_grad_enabled() and any(t.requires_grad for t in ts)

    def info_nce_loss(self, out_1, out_2, temperature=0.07):
        if self._grad(out_1, out_2):
            from mith_train_ops import InfoNceFn
            return InfoNceFn.apply(out_1, out_2, None, temperature)
        return M.info_nce(out_1, out_2, None, temperature)

    def info_nce_loss_bmm(self, out_1, out_2, temperature=0.07):
        a = out_1.permute(1, 0, 2).contiguous()            # [K,N,D] -> [N,K,D]
        b = out_2.permute(1, 0, 2).contiguous()
        if self._grad(a, b):
            from mith_train_ops import InfoNceFn
            return InfoNceFn.apply(a, b, a.shape[1], temperature)
        return M.info_nce(a, b, a.shape[1], temperature)

    def bayesian_loss(self, a, b, label_pair):
        """label_pair = (bank_labels [M,C], batch_labels [B,C]); the [M,B] label_sim is formed inside the kernel."""
        if self._grad(b):
            from mith_train_ops import BayesianLossFn
            return BayesianLossFn.apply(a, b, label_pair[0], label_pair[1])
        return M.bayesian_loss(a, b, label_pair[0], label_pair[1])

    def sq_diff(self, x, y):
        if self._grad(x, y):
            from mith_train_ops import SqDiffSumFn
            return SqDiffSumFn.apply(x, y)
        return M.sq_diff_sum(x, y)

    def quantization_loss_2(self, hash_feature, B):
        return self.sq_diff(hash_feature, B) / (hash_feature.shape[0]) / self.k_bits

    def make_B(self, output_dict):
        ic, it = output_dict['img_cls_hash'].detach(), output_dict['img_tokens_hash'].detach()
        tc, tt = output_dict['txt_cls_hash'].detach(), output_dict['txt_tokens_hash'].detach()
        return M.mith_mix(ic, it, tc, tt, self.args.hyper_lambda)

    def compute_loss(self, output_dict, label, B=None, mixes=None):
        a = self.args
        pair = (self.train_labels.float(), label.to(self.rank).float())
        ic, tc = output_dict['img_cls_hash'], output_dict['txt_cls_hash']
        it, tt = output_dict['img_tokens_hash'], output_dict['txt_tokens_hash']
        Bc, H_i, H_t = mixes if mixes is not None else self.make_B(output_dict)
        if B is not None:
            Bc = B
        if self._grad(ic, it, tc, tt):                     # the hash features that carry gradients (:179-180)
            H_i, H_t = ic * 0.5 + it * 0.5, tc * 0.5 + tt * 0.5
        L = {}
        L['tokens_intra_likelihood'] = a.hyper_tokens_intra * (self.bayesian_loss(self.img_buffer_tokens, it, pair) +
                                                               self.bayesian_loss(self.txt_buffer_tokens, tt, pair))
        L['cls_inter_likelihood'] = a.hyper_cls_inter * (self.bayesian_loss(self.img_buffer_cls, tc, pair) +
                                                         self.bayesian_loss(self.txt_buffer_cls, ic, pair))
        L['quantization'] = a.hyper_quan * (self.quantization_loss_2(H_i, Bc) + self.quantization_loss_2(H_t, Bc))
        # the cls-level InfoNCE is the one term that pairs samples WITH EACH OTHER ([B,B] logits, reference :103-118): with several
        # ranks its inputs are all-gathered (differentiably) so the negatives are the global batch; every other term is a mean of
        # per-sample values against the replicated memory bank, whose rank-local means average to the global one in GradSync
        res_i, res_t = du.gather_loss_inputs(output_dict['res_img_cls'], output_dict['res_txt_cls'])
        L['infoNCE'] = a.hyper_info_nce * (self.info_nce_loss(res_i, res_t) +
                                           a.hyper_alpha * self.info_nce_loss_bmm(output_dict['trans_tokens_i'],
                                                                                  output_dict['trans_tokens_t']))
        # item_1: full gradient to the student (token codes), item_2: a tenth of it to the teacher (cls codes)  (:193-200)
        item_1 = self.sq_diff(ic.detach(), it) + self.sq_diff(tc.detach(), tt)
        item_2 = 0.1 * (self.sq_diff(ic, it.detach()) + self.sq_diff(tc, tt.detach()))
        L['distillation'] = a.hyper_distill * (item_1 + item_2) / ic.shape[0]
        return L

    def train_epoch(self, epoch):
        self.change_state(mode="train")
        self.logger.info(">>>>>> epochs: %d/%d" % (epoch, self.args.epochs))
        for image, text, key_padding_mask, label, index in self.train_loader:
            self.global_step += 1
            image = image.to(self.rank, non_blocking=True)
            text = text.to(self.rank, non_blocking=True)
            key_padding_mask = key_padding_mask.to(self.rank, non_blocking=True)
            output_dict = self.model(image, text, key_padding_mask)
            index = index.to(self.rank)
            codes = [output_dict[k].detach() for k in ('img_cls_hash', 'txt_cls_hash', 'img_tokens_hash', 'txt_tokens_hash')]
            if du.active():
                # every rank's memory bank takes every rank's (index, codes): one fused all-gather per step
                fused, widths = du.fuse_columns(index.view(-1, 1).float(), *codes)
                parts = du.split_columns(du.all_gather_rows(fused, du.row_counts(fused.shape[0], fused.device)), widths)
                index, codes = parts[0].view(-1).long(), parts[1:]
            for buf, rows in zip((self.img_buffer_cls, self.txt_buffer_cls, self.img_buffer_tokens, self.txt_buffer_tokens), codes):
                du.scatter_by_index(buf, index, rows)
            losses = self.compute_loss(output_dict, label)
            loss = sum(losses.values())
            self.optimizer.zero_grad()
            self.backward(loss)
            self.optimizer.step()

    def get_code_MITH(self, data_loader, length: int):
        img_buffer = torch.empty(length, self.args.output_dim, dtype=torch.float).to(self.rank)
        text_buffer = torch.empty(length, self.args.output_dim, dtype=torch.float).to(self.rank)
        seen = []

        def to_device(batch):
            image, text, key_padding_mask, label, index = batch
            out = (image.to(self.rank, non_blocking=True), text.to(self.rank, non_blocking=True),
                   key_padding_mask.to(self.rank, non_blocking=True), index.to(self.rank))
            seen.append(out[3])
            return out

        def work(image, text, key_padding_mask, index):
            od = self.model(image, text, key_padding_mask)
            img_buffer[index, :] = N.sign_codes(od['img_tokens_hash'] + od['img_cls_hash'])
            text_buffer[index, :] = N.sign_codes(od['txt_tokens_hash'] + od['txt_cls_hash'])

        # consecutive batches on two alternating streams (train/base.py); the token-returning trunk has no lock-step pair form
        self._pipelined_batches(data_loader, to_device, work, pair=False)
        self._gather_code_shards(seen, img_buffer, text_buffer)
        return img_buffer, text_buffer, 0

    def _codes_for_eval(self):
        q_img, q_txt, q_t = self.get_code_MITH(self.query_loader, self.args.query_num)
        r_img, r_txt, r_t = self.get_code_MITH(self.retrieval_loader, self.args.retrieval_num)
        return q_img, q_txt, r_img, r_txt, q_t, r_t
