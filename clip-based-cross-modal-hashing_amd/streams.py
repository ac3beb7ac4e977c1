"""Two-tower overlap: the image and the text encoder are independent until the loss (reference model/modelbase.py:105-108
runs them back to back), so each gets its own HIP stream.  On one MI355X this fills the launch gaps and the tails of the
persistent GEMMs of one tower with the other tower's kernels: 6.45 -> 5.65 ms per 256-pair step (bench.py, round 1).

Every side stream first waits for the caller's stream and the caller's stream waits for every side stream at the end, so
tensors may cross freely (same ordering the caching allocator needs)."""
import os

import torch

_streams = {}


def _record(obj, stream):
    """tell the caching allocator that `stream` reads these tensors too (they were allocated on a side stream)"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _record(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record(o, stream)


def overlapped(*fns, inputs_ready=None):
    """Run independent zero-argument callables concurrently, one side stream each; returns their results in order.
    Sequential when CMH_OVERLAP=0, on CPU tensors' behalf (no current CUDA device) or for a single callable.

    inputs_ready (a torch.cuda.Event, or True for "already resident"): PIPELINED form for loops over INDEPENDENT batches (database
    encoding, train/base.py:130-148): the side streams wait only for that event instead of for everything queued on the caller's
    stream, so the towers of batch n + 1 start while batch n's tail - the shorter tower's idle time, the heads, the caller's
    bookkeeping - is still in flight.  The caller's stream still waits for the side streams, results may be used as usual (they are
    registered with the allocator for the caller's stream); the callables must not read anything the caller's stream produced
    after `inputs_ready`."""
    if len(fns) < 2 or os.environ.get("CMH_OVERLAP", "1") == "0" or not torch.cuda.is_available():
        return tuple(f() for f in fns)
    dev = torch.cuda.current_device()
    pool = _streams.setdefault(dev, [])
    while len(pool) < len(fns):
        pool.append(torch.cuda.Stream(device=dev))
    cur = torch.cuda.current_stream(dev)
    out = []
    for s, f in zip(pool, fns):
        if inputs_ready is None:
            s.wait_stream(cur)
        elif inputs_ready is not True:
            s.wait_event(inputs_ready)
        with torch.cuda.stream(s):
            out.append(f())
    for s in pool[:len(fns)]:
        cur.wait_stream(s)
    _record(out, cur)      # the results were allocated on the side streams and are read on the caller's: tell the allocator (the caller
                           # may itself be one of several alternating streams, whose next overlapped() call does not wait for this one)
    return tuple(out)


class AlternatingStreams:
    """A loop over INDEPENDENT batches (database encoding: train/base.py:130-148 of the reference walks the loader batch by batch)
    on two HIP streams in turn.  One batch's chain of launches leaves the GPU idle in places - launch gaps, the single-wave tails of
    the persistent GEMMs, the bandwidth-bound LayerNorm / attention launches between them; the next batch's chain on the other
    stream fills them.  With the lock-step pair path (CLIP.encode_pair: layer i of both towers shares its launches) as the chain,
    this is the fastest form of the encode loop measured (bench.py --towers pair2: -3 % against one stream per tower).

    run(fn, inputs, ready): fn() executes on the next stream once `ready` (an event on the caller's stream: the batch's inputs are
    there) has passed; `inputs` are registered with the allocator for that stream.  Whatever fn writes (code buffers) belongs to
    the side streams until join(), which makes the caller's stream wait for both."""

    def __init__(self, device=None):
        self.device = torch.cuda.current_device() if device is None else device
        self.enabled = torch.cuda.is_available() and os.environ.get("CMH_OVERLAP", "1") != "0"
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(2)] if self.enabled else []
        self.turn = 0

    def run(self, fn, inputs=(), ready=None):
        if not self.enabled:
            return fn()
        s = self.streams[self.turn & 1]
        self.turn += 1
        if ready is not None:
            s.wait_event(ready)
        elif self.turn <= 2:                       # first use of each stream: everything the caller queued so far
            s.wait_stream(torch.cuda.current_stream(self.device))
        _record(inputs, s)
        with torch.cuda.stream(s):
            return fn()

    def join(self):
        if self.enabled:
            cur = torch.cuda.current_stream(self.device)
            for s in self.streams:
                cur.wait_stream(s)
