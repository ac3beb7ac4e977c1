"""Trainer base with the reference's hooks (train/base.py:14-349): _init_dataset/_init_model/
_init_writer/run/change_state/get_code*/train/valid/test/save_model/save_mat and the attributes
subclasses read (model, optimizer, loaders, train_labels, query_labels, retrieval_labels, rank,
logger, global_step).

Differences from upstream, all deliberate:
  * `rank` really is the device index passed by main.py (upstream drops it and lands on cuda:1, F7);
  * valid() dispatches get_code_DNPH on 'DNPH' (upstream tests 'DNPH-TOMM', which never matches);
  * code generation runs under no_grad (upstream builds and discards an autograd graph);
  * codes are sign()/argmax'd by libcmh kernels and ranked on the GPU (utils/calc_utils.py).
Real datasets (`--data-dir` with index.mat / caption.mat|txt / label.mat, dataset/dataloader.py) go through the GPU-side
input pipeline of dataset/base.py; `--dataset synthetic` exercises the same contract without files."""
import os
import time

import torch
from torch.utils.data import DataLoader

import cmh_native as N
import dist_utils as du
from streams import AlternatingStreams
from utils import get_logger, get_summary_writer
from utils.calc_utils import calc_map_k_matrix as calc_map_k

NCLASS = {'flickr': 24, 'coco': 80, 'nuswide': 21, 'iapr': 291, 'synthetic': 24}


class TrainBase(object):

    def __init__(self, args, rank=0):
        self.args = args
        os.makedirs(args.save_dir, exist_ok=True)
        self._init_writer()
        self.logger.info(self.args)
        self.rank = getattr(args, "rank", rank)   # GPU ID
        self._init_dataset()
        self._init_model()
        if du.active():          # replicas start from rank 0's weights (heads and loss parameters are drawn at random)
            du.broadcast_modules_([m for m in vars(self).values() if isinstance(m, torch.nn.Module)])
        self.global_step = 0
        self.max_mapi2t = 0
        self.max_mapt2i = 0
        self.best_epoch_i = 0
        self.best_epoch_t = 0

    def _init_dataset(self):
        self.logger.info("init dataset.")
        self.logger.info(f"Using {self.args.dataset} dataset.")
        if self.args.dataset not in NCLASS:
            raise ValueError("Unknown dataset")
        self.args.nclass = NCLASS[self.args.dataset]
        if self.args.dataset == 'synthetic':
            if self.args.method == 'MITH':      # MITH batches carry a key_padding_mask (train/MITH/data.py)
                from dataset.synthetic import dataloader_mith as dataloader
            else:
                from dataset.synthetic import dataloader
            train_data, query_data, retrieval_data = dataloader(
                total=self.args.synthetic_size, nclass=self.args.nclass, maxWords=self.args.max_words,
                imageResolution=self.args.resolution, query_num=self.args.query_num,
                train_num=self.args.train_num, seed=self.args.seed)
        else:
            # reference train/base.py:39-88: <data dir>/index.mat, caption.mat (caption.txt for nuswide), label.mat.  Upstream
            # hard-codes the directory ('YOUR-FLIE-DIR'); here it is --data-dir.  Items carry raw pixels and caption strings,
            # the batch is tokenised natively and the image transform runs on the GPU (dataset/base.py).
            data_dir = self.args.data_dir
            if not data_dir or not os.path.isdir(data_dir):
                raise FileNotFoundError(f"--data-dir {data_dir!r}: directory with index.mat / caption.mat|txt / label.mat expected")
            self.index_file = os.path.join(data_dir, "index.mat")
            self.caption_file = os.path.join(data_dir, "caption.txt" if 'nuswide' in self.args.dataset else "caption.mat")
            self.label_file = os.path.join(data_dir, "label.mat")
            if self.args.method == 'MITH':
                from train.MITH.data import generate_dataset as dataloader
            else:
                from dataset.dataloader import dataloader
            train_data, query_data, retrieval_data = dataloader(
                captionFile=self.caption_file, indexFile=self.index_file, labelFile=self.label_file,
                maxWords=self.args.max_words, imageResolution=self.args.resolution, query_num=self.args.query_num,
                train_num=self.args.train_num, seed=self.args.seed)
        self.train_labels = train_data.get_all_label().to(self.rank)
        self.query_labels = query_data.get_all_label()
        self.retrieval_labels = retrieval_data.get_all_label()
        self.args.retrieval_num = len(self.retrieval_labels)
        self.args.query_num = len(self.query_labels)
        self.logger.info(f"query shape: {self.query_labels.shape}")
        self.logger.info(f"retrieval shape: {self.retrieval_labels.shape}")
        self.train_sampler, self.train_loader = self._loader(train_data, train=True)
        _, self.query_loader = self._loader(query_data, train=False)
        _, self.retrieval_loader = self._loader(retrieval_data, train=False)

    def _loader(self, data, train):
        """Upstream: DataLoader(batch_size, num_workers, pin_memory=True, shuffle=True) for all three sets (train/base.py:90-113).
        With one process per GPU each rank draws its share of every epoch from a DistributedSampler (the short tail is
        padded by repeats, so all ranks run the same number of steps); evaluation sets are cut the same way, unshuffled."""
        sampler = None
        if du.active():
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(data, shuffle=train, seed=int(getattr(self.args, "seed", 0) or 0))
        kw = dict(batch_size=self.args.batch_size, num_workers=self.args.num_workers, sampler=sampler, shuffle=sampler is None)
        if self.args.dataset == 'synthetic':
            return sampler, DataLoader(dataset=data, pin_memory=True, **kw)
        from dataset.base import DeviceLoader
        return sampler, DeviceLoader(data, self.rank, **kw)

    def _init_model(self):
        self.model = None
        self.model_ddp = None

    def _init_writer(self):
        r = du.dist_rank()
        self.is_main = r == 0                     # checkpoints, .mat files and the tensorboard directory come from rank 0 only
        tag = "" if self.is_main else f".rank{r}"
        self.logger = get_logger(os.path.join(self.args.save_dir, ("train" if self.args.is_train else "test") + tag + ".log"))
        self.writer = get_summary_writer(os.path.join(self.args.save_dir, "tensorboard" + tag))

    def run(self):
        if self.args.is_train:
            self.train()
        else:
            self.test()

    def change_state(self, mode):
        if mode == "train":
            self.model.train()
        elif mode == "valid":
            self.model.eval()

    # ---- code generation (train/base.py:130-223) ---------------------------------------------------
    def _pipelined_batches(self, data_loader, to_device, work, pair=True):
        """The evaluation loops walk a loader batch by batch (reference train/base.py:130-148, train/TwDH/hash_train.py:165-204,
        train/MITH/hash_train.py get_code): the batches are independent, so consecutive ones alternate between two HIP streams
        (streams.AlternatingStreams) - one batch's launch gaps and bandwidth-bound kernels run beside the other's GEMMs - and, with
        `pair`, a batch's two towers run in lock-step (CLIP.prefetch_pair: layer i of both shares its launches, the method model's
        encode_image / encode_text calls then pick the features up).  Same values, bit for bit, as the plain loop
        (CMH_OVERLAP=0 CMH_PAIR=0).  to_device(batch) -> tuple of device tensors (image, text, ...) moved on the CALLER's stream;
        work(*tensors) runs on a side stream and writes its results into buffers the caller reads after the loop."""
        clip = getattr(self.model, "clip", None)
        use_pair = pair and clip is not None and hasattr(clip, "prefetch_pair") and os.environ.get("CMH_PAIR", "1") != "0"
        pipe = AlternatingStreams(torch.device("cuda", self.rank) if isinstance(self.rank, int) else self.rank)
        # Round 5: two consecutive loader batches share ONE run of the towers (CLIP.prefetch_pairs -> cmh_clip_encode_pair2): a launch's
        # fixed costs - ramp, first stage, last epilogue: ~10 us of a 50 us GEMM launch at batch 256 - are paid once per 512 pairs
        # (profiles/r05_k: -10 % GEMM time per pair).  work() still runs per loader batch; same values, bit for bit.  CMH_COALESCE=1: off.
        group = 2 if (use_pair and pipe.enabled and hasattr(clip, "prefetch_pairs") and os.environ.get("CMH_COALESCE", "2") != "1") else 1

        def dispatch(held):
            ready = None
            if pipe.enabled:
                ready = torch.cuda.Event()
                ready.record()

            def run(held=held):
                ok = [use_pair and t[0].shape[0] == t[1].shape[0] for t in held]
                try:
                    if len(held) == 2 and all(ok) and held[0][0].shape[1:] == held[1][0].shape[1:] and held[0][1].shape[1:] == held[1][1].shape[1:]:
                        clip.prefetch_pairs([(t[0], t[1]) for t in held])
                        for t in held:
                            work(*t)
                    else:
                        for t, o in zip(held, ok):
                            if o:
                                clip.prefetch_pair(t[0], t[1])
                            work(*t)
                            if use_pair:
                                clip.drop_pair_stash()
                finally:
                    if use_pair:
                        clip.drop_pair_stash()      # a work() that took one side only (or raised) must not pin the batch

            pipe.run(run, tuple(x for t in held for x in t), ready)

        with torch.no_grad():
            held = []
            for batch in data_loader:
                # only batches that are ALREADY on the device wait for a partner (dataset/base.py::DeviceLoader's cached epochs hand
                # those out): a batch still on the host has its 154 MB of pixels to copy first, and holding the towers back until two
                # copies are through costs more (-5 % measured) than the shared launches return
                resident = isinstance(batch[0], torch.Tensor) and batch[0].is_cuda
                held.append(tuple(to_device(batch)))
                if len(held) == group or not resident:
                    dispatch(held)
                    held = []
            if held:
                dispatch(held)
        pipe.join()

    def _code_loop(self, data_loader, length, encode):
        img_buffer = torch.empty(length, self.args.output_dim, dtype=torch.float).to(self.rank)
        text_buffer = torch.empty(length, self.args.output_dim, dtype=torch.float).to(self.rank)
        seen = []

        def to_device(batch):
            image, text, index = batch[0], batch[1], batch[-1]
            out = (image.to(self.rank, non_blocking=True), text.to(self.rank, non_blocking=True), index.to(self.rank))
            seen.append(out[2])
            return out

        def work(image, text, index):
            image_hash, text_hash = encode(image, text, None)
            img_buffer[index, :] = image_hash
            text_buffer[index, :] = text_hash

        start_encoder_time = time.time()
        self._pipelined_batches(data_loader, to_device, work)
        encoder_time = time.time() - start_encoder_time
        self._gather_code_shards(seen, img_buffer, text_buffer)
        return img_buffer, text_buffer, encoder_time

    def _gather_code_shards(self, seen, *buffers):
        """With one process per GPU every rank has encoded its share of the set (rows `seen` of each buffer): ONE fused all-gather
        fills the rest, so all ranks hold the whole code matrices before ranking.  What travels is PACKED (SURVEY 8e "all-gather
        packed DB codes"): per item an int32 position and, per code matrix, the two bit planes of cmh_pack_codes (sign, non-zero) -
        4 + 2 * 2 * ceil(K/32) * 4 bytes for an (image, text) pair instead of 4 + 2 * K * 4 bytes of floats (NUS-WIDE, 128 bit:
        6.9 MB instead of 196 MB); every rank then restores the reference's float buffers (train/base.py:130-148: save_mat and
        calc_map_k take them) with cmh_unpack_codes.  Codes are exactly -1 / 0 / +1 (sign, argmax), so nothing is lost."""
        if not du.active() or not seen:
            return
        mine = torch.cat(seen)
        cols, widths = [mine.to(torch.int32).unsqueeze(1)], [1]
        for b in buffers:
            sp, nz = N.pack_codes(b[mine])                          # validates the -1 / 0 / +1 domain (one flag read per matrix)
            cols += [sp, nz]
            widths += [sp.shape[1], nz.shape[1]]
        fused = torch.cat(cols, dim=1).contiguous()
        parts = torch.split(du.all_gather_rows(fused, du.row_counts(mine.numel(), fused.device)), widths, dim=1)
        pos = parts[0].reshape(-1).long()
        for i, buf in enumerate(buffers):
            buf[pos] = N.unpack_codes(parts[1 + 2 * i], parts[2 + 2 * i], buf.shape[1])

    def get_code(self, data_loader, length: int):
        return self._code_loop(data_loader, length, lambda i, t, b: (
            N.sign_codes(self.model.encode_image(i)), N.sign_codes(self.model.encode_text(t))))

    def make_hash_code_DCHMT(self, code) -> torch.Tensor:
        """argmax over each pair; index 0 -> -1, 1 -> +1 (train/base.py:150-158)."""
        p = torch.cat(code, dim=-1) if isinstance(code, (list, tuple)) else code
        return N.pair_argmax_codes(p)

    def get_code_DCHMT(self, data_loader, length: int):
        return self._code_loop(data_loader, length, lambda i, t, b: (
            self.make_hash_code_DCHMT(self.model.encode_image(i)), self.make_hash_code_DCHMT(self.model.encode_text(t))))

    def get_code_DNPH(self, data_loader, length: int):
        return self._code_loop(data_loader, length, lambda i, t, b: (
            N.sign_codes(self.model.encode_image(i)[0]), N.sign_codes(self.model.encode_text(t)[0])))

    def backward(self, loss, *loss_modules):
        """loss.backward(); with one process per GPU also the gradient means over the ranks, each tower's all-reduce queued from
        inside the backward pass as soon as its last gradient is written (dist_utils.GradSync)."""
        if du.active() and getattr(self, "_grad_sync", None) is None:
            self._grad_sync = du.GradSync.for_model(self.model, *loss_modules)
        loss.backward()
        if getattr(self, "_grad_sync", None) is not None:
            self._grad_sync.finish()

    def save_model(self, epoch):
        if not getattr(self, "is_main", True):
            return
        torch.save(self.model.state_dict(), os.path.join(self.args.save_dir, "model-" + str(epoch) + ".pth"))
        self.logger.info("save mode to {}".format(os.path.join(self.args.save_dir, "model-" + str(epoch) + ".pth")))

    def train_epoch(self, epoch):
        raise NotImplementedError("Function of 'train' doesn't implement.")

    def train(self):
        self.logger.info("Start train.")
        for epoch in range(self.args.epochs):
            if getattr(self, "train_sampler", None) is not None:
                self.train_sampler.set_epoch(epoch)
            self.train_epoch(epoch)
            if du.active():     # replicas must hold identical weights after every epoch: log a checksum per rank
                with torch.no_grad():
                    total = sum(p.double().sum() for p in self.model.parameters())
                self.logger.info(f">>>>>> [{epoch}/{self.args.epochs}] replica checksum: {float(total)!r}")
            self.valid(epoch)
            self.save_model(epoch)
        self.logger.info(
            f">>>>>>> FINISHED >>>>>> Best epoch, I-T: {self.best_epoch_i}, mAP: {self.max_mapi2t}, T-I: {self.best_epoch_t}, mAP: {self.max_mapt2i}")

    def _codes_for_eval(self):
        if self.args.method == 'DCHMT':
            gc = self.get_code_DCHMT
        elif self.args.method in ('DNPH', 'DNPH-TOMM'):
            gc = self.get_code_DNPH
        else:
            gc = self.get_code
        q_img, q_txt, q_t = gc(self.query_loader, self.args.query_num)
        r_img, r_txt, r_t = gc(self.retrieval_loader, self.args.retrieval_num)
        return q_img, q_txt, r_img, r_txt, q_t, r_t

    def _map(self, query_codes, retrieval_codes, k=None):
        """calc_map_k for one direction.  With one process per GPU every rank holds the whole code matrices (_gather_code_shards),
        ranks its own contiguous share of the QUERIES against the whole database, and the per-query APs are gathered in query
        order and summed in f32 exactly like the kernel's own mean (reference utils/calc_utils.py:37-38), so the value is the
        single-GPU one bit for bit (reference call sites train/base.py:259-262, :299-302)."""
        if not du.active():
            return calc_map_k(query_codes, retrieval_codes, self.query_labels, self.retrieval_labels, k, self.rank)
        n_query = query_codes.shape[0]
        lo, hi = du.query_shard(n_query)
        if hi > lo:
            _, ap = calc_map_k(query_codes[lo:hi], retrieval_codes, self.query_labels[lo:hi], self.retrieval_labels, k, self.rank,
                               return_ap=True)
        else:
            ap = torch.empty(0, dtype=torch.float32, device=query_codes.device)
        return du.mean_in_query_order(du.gather_query_sharded_ap(ap, n_query))

    def _four_maps(self, query_img, query_txt, retrieval_img, retrieval_txt, k=None):
        mAPi2t = self._map(query_img, retrieval_txt, k)
        mAPt2i = self._map(query_txt, retrieval_img, k)
        mAPi2i = self._map(query_img, retrieval_img, k)
        mAPt2t = self._map(query_txt, retrieval_txt, k)
        return mAPi2t, mAPt2i, mAPi2i, mAPt2t

    def loss_inputs(self, *blocks):
        """The per-rank loss inputs as global-batch tensors: ONE fused, differentiable all-gather per step (dist_utils.
        gather_loss_inputs); the pairwise terms of the reference's losses are O(B^2) in the GLOBAL batch
        (train/DSPH/loss.py:42-66, train/DCHMT/hash_train.py:82-114).  One rank: the blocks themselves."""
        return du.gather_loss_inputs(*blocks)

    def valid(self, epoch):
        self.logger.info("Valid.")
        self.change_state(mode="valid")
        query_img, query_txt, retrieval_img, retrieval_txt, q_encoder_time, r_encoder_time = self._codes_for_eval()
        mAPi2t, mAPt2i, mAPi2i, mAPt2t = self._four_maps(query_img, query_txt, retrieval_img, retrieval_txt)
        if self.max_mapi2t < mAPi2t:
            self.best_epoch_i = epoch
            self.save_mat(query_img, query_txt, retrieval_img, retrieval_txt, mode_name="i2t")
        self.max_mapi2t = max(self.max_mapi2t, mAPi2t)
        if self.max_mapt2i < mAPt2i:
            self.best_epoch_t = epoch
            self.save_mat(query_img, query_txt, retrieval_img, retrieval_txt, mode_name="t2i")
        self.max_mapt2i = max(self.max_mapt2i, mAPt2i)
        self.logger.info(
            f">>>>>> [{epoch}/{self.args.epochs}], MAP(i->t): {mAPi2t}, MAP(t->i): {mAPt2i}, MAP(t->t): {mAPt2t}, MAP(i->i): {mAPi2i}, \
                            MAX MAP(i->t): {self.max_mapi2t}, MAX MAP(t->i): {self.max_mapt2i}, query_encoder_time: {q_encoder_time}, retrieval_encoder_time: {r_encoder_time}")
        return mAPi2t, mAPt2i, mAPi2i, mAPt2t

    def test(self, mode_name="i2t"):
        if self.args.pretrained == "":
            raise RuntimeError("test step must load a model! please set the --pretrained argument.")
        self.change_state(mode="valid")
        query_img, query_txt, retrieval_img, retrieval_txt, _, _ = self._codes_for_eval()
        mAPi2t, mAPt2i, mAPi2i, mAPt2t = self._four_maps(query_img, query_txt, retrieval_img, retrieval_txt)
        self.max_mapt2i = max(self.max_mapt2i, mAPt2i)
        self.logger.info(f">>>>>> MAP(i->t): {mAPi2t}, MAP(t->i): {mAPt2i}, MAP(t->t): {mAPt2t}, MAP(i->i): {mAPi2i}")
        self.save_mat(query_img, query_txt, retrieval_img, retrieval_txt, mode_name=mode_name)
        self.logger.info(">>>>>> save all data!")

    def compute_loss(self):
        raise NotImplementedError("Function of 'compute_loss' doesn't implement.")

    def save_mat(self, query_img, query_txt, retrieval_img, retrieval_txt, mode_name="i2t"):
        """PR_cruve/<bits>-ours-<dataset>-<mode>.mat with q_img q_txt r_img r_txt q_l r_l (train/base.py:328-349)."""
        if not getattr(self.args, "save_mat", True) or not getattr(self, "is_main", True):
            return
        import scipy.io as scio
        save_dir = os.path.join(self.args.save_dir, "PR_cruve")
        os.makedirs(save_dir, exist_ok=True)
        result_dict = {
            'q_img': query_img.cpu().detach().numpy(), 'q_txt': query_txt.cpu().detach().numpy(),
            'r_img': retrieval_img.cpu().detach().numpy(), 'r_txt': retrieval_txt.cpu().detach().numpy(),
            'q_l': self.query_labels.numpy(), 'r_l': self.retrieval_labels.numpy(),
        }
        scio.savemat(os.path.join(save_dir, str(self.args.output_dim) + "-ours-" + self.args.dataset + "-" + mode_name + ".mat"), result_dict)
        self.logger.info(f">>>>>> save best {mode_name} data!")
