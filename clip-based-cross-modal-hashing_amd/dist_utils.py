"""Multi-GPU plumbing for the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed code at all; north_star adds exactly one exchange step:
an ALL-GATHER of per-rank code blocks (hash outputs / sign codes, plus labels and dataset indices) so
that (i) pairwise losses see the global batch and (ii) every rank holds the whole retrieval database
(3 MB packed at NUS-WIDE scale) while QUERIES are sharded.  Messages are tens of KiB: latency-bound,
so each step issues ONE fused all-gather of a [B_local, width] block, never one per tensor.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def forced() -> bool:
    """CMH_FORCE_DIST=1: join a process group and run EVERY collective of the path even when the group has one rank.  A group of
    one is what a one-GPU box can give RCCL: communicator creation bound to the device, all_gather_into_tensor / all_reduce /
    broadcast on device tensors, the asynchronous in-place buckets of GradSync and their stream hand-over all execute for real
    (tests/test_gpu_rccl_one_rank.py); what it cannot show is a ring over xGMI or a rank-dependent bug."""
    return os.environ.get("CMH_FORCE_DIST", "0") == "1"


def active() -> bool:
    """Do the collectives of the path run in this process?  (a group of more than one rank, or a forced group of one)"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init_from_env(backend: str | None = None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun).  -> (rank, world, local)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("CMH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":      # bind the communicator to this rank's GPU up front (one process per GPU)
            ndev = max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local % ndev)
            kw["device_id"] = torch.device("cuda", local % ndev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def dist_rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n items; the first n % world ranks get one extra item."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def fuse_columns(*blocks: torch.Tensor) -> tuple[torch.Tensor, list[int]]:
    """[B, w_i] blocks -> one [B, sum w_i] f32 block (one collective instead of len(blocks))."""
    widths = [b.shape[1] for b in blocks]
    return torch.cat([b.float() for b in blocks], dim=1).contiguous(), widths


def split_columns(fused: torch.Tensor, widths: list[int]) -> list[torch.Tensor]:
    return list(torch.split(fused, widths, dim=1))


def all_gather_rows(block: torch.Tensor, counts: list[int] | None = None) -> torch.Tensor:
    """Concatenate the [B_r, w] blocks of all ranks in rank order.  Equal B_r -> one all_gather_into_tensor
    (RCCL ring over xGMI); ragged -> pad to the max, gather, and strip (counts = rows per rank)."""
    if not active():
        return block
    world = dist.get_world_size()
    block = block.contiguous()
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out, block)
        return out
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    pad[:block.shape[0]] = block
    out = torch.empty((world * mx,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(world)], dim=0)


class _GatherRowsFn(torch.autograd.Function):
    """all_gather_rows with a backward: every rank computes the SAME loss on the gathered rows, so the gradient of that loss with
    respect to rank r's block is rows [r*B, (r+1)*B) of the incoming gradient.  It is handed back times `world`: the parameter
    gradients are then averaged over the ranks (GradSync / allreduce_mean_), and (1/world) * sum_r world * dL/d(rows of r)
    is the gradient a single GPU would compute for the whole global batch."""

    @staticmethod
    def forward(ctx, block):
        ctx.rows, ctx.rank, ctx.world = block.shape[0], dist_rank(), world_size()
        return all_gather_rows(block)

    @staticmethod
    def backward(ctx, grad):
        lo = ctx.rank * ctx.rows
        return grad[lo:lo + ctx.rows] * float(ctx.world)


def gather_loss_inputs(*blocks: torch.Tensor) -> list[torch.Tensor]:
    """The exchange step of a training iteration (north_star "all-gather of hash codes ... for the pairwise loss"):
    the per-rank [B_local, w_i] loss inputs (hash outputs, labels, ...) travel as ONE fused [B_local, sum w_i] all-gather and
    come back as the global-batch tensors [world * B_local, w_i], rank-major, on every rank.  Differentiable (see
    _GatherRowsFn); blocks that need no gradient (labels) ride along in the same message.  Every rank must contribute the same
    number of rows (the trainers' DistributedSampler pads the epoch so that they do).  One rank: returned unchanged."""
    if not active():
        return list(blocks)
    widths = [b.shape[1] for b in blocks]
    fused = torch.cat([b.float() for b in blocks], dim=1).contiguous()
    if fused.requires_grad:
        whole = _GatherRowsFn.apply(fused)
    else:
        whole = all_gather_rows(fused)
    return list(torch.split(whole, widths, dim=1))


def broadcast_tensor_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """t on every rank := rank `src`'s (a random draw that a single-GPU run would make once for the whole batch)."""
    if active():
        dist.broadcast(t, src=src)
    return t


def query_shard(n_query: int) -> tuple[int, int]:
    """This rank's contiguous share [lo, hi) of the queries of an evaluation."""
    return shard_range(n_query, dist_rank(), world_size())


def row_counts(n_local: int, device) -> list[int] | None:
    """Rows every rank contributes to a ragged all_gather_rows (the last batch of an epoch may differ between ranks)."""
    if not active():
        return None
    mine = torch.tensor([n_local], dtype=torch.int64, device=device)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return [int(t.item()) for t in every]


def gather_query_sharded_ap(ap_local: torch.Tensor, n_query: int) -> torch.Tensor:
    """Per-query APs computed on query shards -> the full [Q] vector in QUERY ORDER on every rank, so the final
    mean is accumulated in the same order as a single-GPU run (the reference sums in query order)."""
    world = world_size()
    if not active():
        return ap_local
    counts = [shard_range(n_query, r, world)[1] - shard_range(n_query, r, world)[0] for r in range(world)]
    return all_gather_rows(ap_local.reshape(-1, 1), counts).reshape(-1)


def mean_in_query_order(ap: torch.Tensor) -> torch.Tensor:
    """f32 running sum in query order / Q (utils/calc_utils.py:37-38 `map += AP; map / num_query`).  On the GPU this is the very
    kernel that cmh_hamming_map appends to a ranking (cmh_map_mean: one thread adds the APs in order), so the mean of gathered
    per-query APs is the single-GPU value bit for bit; CPU tensors (the gloo tests) take the same sum as a host loop."""
    if ap.is_cuda:
        import cmh_native as N
        return N.map_mean(ap.detach())
    acc = torch.zeros((), dtype=torch.float32)
    for v in ap.detach().float():
        acc = acc + v
    return acc / ap.numel()


def scatter_by_index(buffer: torch.Tensor, index: torch.Tensor, rows: torch.Tensor) -> None:
    """buffer[index] = rows for the gathered (index, rows) of all ranks — the reference's code buffers and the
    MITH memory bank are indexed by dataset position (train/base.py:145-146, train/MITH/hash_train.py:72-78)."""
    buffer[index.long()] = rows.to(buffer.dtype)


def broadcast_modules_(modules, src: int = 0) -> None:
    """Parameters and buffers of `modules` on every rank := rank `src`'s, in flat per-dtype messages."""
    if not active():
        return
    seen, by_dtype = set(), {}
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if id(t) not in seen:
                seen.add(id(t))
                by_dtype.setdefault(t.dtype, []).append(t)
    for group in by_dtype.values():
        with torch.no_grad():
            flat = torch.cat([t.reshape(-1) for t in group])
            dist.broadcast(flat, src=src)
            off = 0
            for t in group:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


def allreduce_mean_(tensors: list[torch.Tensor], bucket_bytes: int = 256 << 20) -> None:
    """Data-parallel gradient synchronisation: average `tensors` over the ranks in place.  Gradients are
    packed into flat buckets of ~bucket_bytes so that a step is a handful of large ring all-reduces (xGMI rings are per-link
    bound: few big messages, not 302 small ones), one bucket in flight while the next is being packed."""
    world = world_size()
    if not active() or not tensors:
        return
    buckets, cur, cur_bytes = [], [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    pending = []
    for group in buckets:
        flat = torch.cat([t.reshape(-1) for t in group])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, group))
    for work, flat, group in pending:
        work.wait()
        flat.div_(world)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class GradSync:
    """Gradient means over the ranks, queued from inside the backward pass ("bucketed ... overlapped with backward").

    Two routes, one `finish()`:
      * BUCKETS of the native towers.  cmh_vit_backward / cmh_text_backward run in parts (model/base/train_ops.py: the head + the
        last blocks first, the embeddings last); all gradients of a tower call live in ONE flat buffer in completion order, and the
        moment a part returns its slice of that buffer is all-reduced IN PLACE (asynchronously: the first message of a step leaves
        after a third of the first tower's backward).  The parameters' .grad are views of the same buffer, so nothing is packed and
        nothing is copied back; `finish()` only waits and divides.
      * HOOKS for everything else (hash heads, loss parameters, any module that does not go through the native bridges): the
        parameters are cut into groups, a post-accumulate hook on every parameter counts its group down, and a complete group is
        packed into one flat message (these groups are a few hundred KB).  Which parameters receive a gradient is learnt from the
        previous step (the first step sends everything from `finish()`); a gradient that shows up after its group was sent goes out
        in `finish()`.

    Same sums and the same division as allreduce_mean_ over the same gradients.  Contract: in a given step every rank produces
    gradients for the same parameters, one backward pass per `finish()`, gradients cleared (set to None) between steps."""

    def __init__(self, groups):
        self.world = world_size()
        self.on = active()                                    # collectives run (more than one rank, or CMH_FORCE_DIST=1)
        self.groups = [[p for p in g if p.requires_grad] for g in groups]
        self.groups = [g for g in self.groups if g]
        self._group_of = {id(p): gi for gi, g in enumerate(self.groups) for p in g}
        self._expected = [None] * len(self.groups)            # ids of the parameters that had a gradient last step
        self._arrived = [[] for _ in self.groups]
        self._sent = [False] * len(self.groups)
        self._late, self._pending, self._handles = [], [], []
        self._buckets, self._via_bucket = [], set()           # in-place messages of this step; ids of the parameters they cover
        self._deferred = []                                   # CMH_GRADSYNC_DEFER=1: buckets held back until finish()
        self.bucket_log = []                                  # (elements, parameters) of every in-place message of the last step
        if self.on:
            for g in self.groups:
                for p in g:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
            self._install_sink()

    def _install_sink(self):
        try:
            from model.base import train_ops
        except ImportError:                                    # host-only use (tests of the hook route)
            self._train_ops = None
            return
        self._train_ops = train_ops
        train_ops.BUCKET_SINK = self._on_bucket

    @staticmethod
    def for_model(model, *extra_modules):
        """[text tower], [image tower], [everything else] of a Baseclip-style model (+ loss modules holding parameters)."""
        clip = getattr(model, "clip", None)
        visual = list(clip.visual.parameters()) if clip is not None and hasattr(clip, "visual") else []
        vis_ids = {id(p) for p in visual}
        text = [p for p in clip.parameters() if id(p) not in vis_ids] if clip is not None else []
        seen = vis_ids | {id(p) for p in text}
        rest = [p for m in (model,) + extra_modules for p in m.parameters() if id(p) not in seen]
        return GradSync([text, visual, rest])

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        if getattr(self, "_train_ops", None) is not None and self._train_ops.BUCKET_SINK == self._on_bucket:
            self._train_ops.BUCKET_SINK = None

    def _on_bucket(self, flat, params, views):
        """a part of a tower's backward has written its last gradient: all-reduce its slice of the flat buffer where it lies"""
        if not self.on:
            return
        # remember WHERE each gradient lies, not the view objects: autograd adopts a returned gradient as p.grad only while nobody
        # else holds a reference to it (otherwise it clones it)
        spans = [(v.storage_offset() - flat.storage_offset(), v.numel()) for v in views]
        if os.environ.get("CMH_GRADSYNC_DEFER") == "1":       # measurement only (tools/gradsync_overlap.sh): every bucket leaves from finish()
            self._deferred.append((flat, params, spans))
        else:
            self._buckets.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, params, spans))
        self._via_bucket.update(id(p) for p in params)

    def _send(self, params):
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        self._pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, params))

    def _on_grad(self, p):
        if id(p) in self._via_bucket:          # already travelling inside its tower's bucket
            return
        gi = self._group_of[id(p)]
        if self._sent[gi]:
            self._late.append(p)
            return
        self._arrived[gi].append(p)
        exp = self._expected[gi]
        if exp is not None and len(self._arrived[gi]) == len(exp) and {id(q) for q in self._arrived[gi]} == exp:
            self._send(self._arrived[gi])
            self._sent[gi] = True

    def finish(self):
        if not self.on:
            return
        for gi in range(len(self.groups)):
            if not self._sent[gi] and self._arrived[gi]:
                self._send(self._arrived[gi])
        if self._late:
            self._send(self._late)
        for flat, params, spans in self._deferred:
            self._buckets.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, params, spans))
        self._deferred = []
        on_gpu = torch.cuda.is_available()
        self.bucket_log = []
        for work, flat, params, spans in self._buckets:
            work.wait()
            flat.div_(self.world)
            self.bucket_log.append((flat.numel(), len(params)))
            for p, (off, n) in zip(params, spans):
                # autograd adopts the view itself as p.grad; if it made its own tensor instead (a gradient accumulated into an
                # existing one, a layout it did not like), that tensor holds this rank's values and takes the mean from the buffer
                if p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + 4 * off:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
        for work, flat, params in self._pending:
            work.wait()
            if on_gpu and flat.is_cuda:
                flat.record_stream(torch.cuda.current_stream(flat.device))
            flat.div_(self.world)
            off = 0
            for p in params:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        late_ids = {id(p) for p in self._late}
        for gi in range(len(self.groups)):
            got = {id(p) for p in self._arrived[gi]} | {i for i in late_ids if self._group_of[i] == gi}
            self._expected[gi] = got if got else None
        self._arrived = [[] for _ in self.groups]
        self._sent = [False] * len(self.groups)
        self._late, self._pending = [], []
        self._buckets, self._via_bucket = [], set()
