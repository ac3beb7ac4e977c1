#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json metric: image+text pairs/s encoded+hashed per GPU).

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
  encode_image (ViT-B/32, 224x224) + encode_text (77 tokens) -> LinearHash heads (tanh) -> sign() codes ->
  bit-packed codes -> [N>1: ONE fused RCCL all-gather of the hash outputs + labels] -> DSPH HyP loss (forward).
Workload = BASELINE.json configs[1]: "DSPH --dataset flickr25k --output-dim 64, ViT-B/32 bf16, 1xMI355X,
batch=256" (per-GPU batch fixed as N grows -> weak scaling).  Random-init weights, synthetic data.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     dominant kernel (the MFMA GEMM): algorithmic FLOPs / HIP-event launch durations, live, over
               the timed region (cmh_prof_gemm_*), vs the dense MFMA peak of the dtype.
  cpu_baseline the oracle (numpy fp32 restatement of the reference path) timed on the host cores, rank 0, N=1.
  map_eval     secondary metric: wall-clock of the 4 calc_map_k directions (64-bit, MIRFlickr scale), codes
               resident; not part of `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "clip-based-cross-modal-hashing_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}      # /opt/skills/guides/MI355X_MICROARCH.md, dense
FLOP_IMG = 8.818e9                                  # BASELINE.md §3 (fwd, per image)
VITB32 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
              context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12)


def text_flops(L):
    d, layers = 512, 12
    per_tok = layers * (2 * d * 3 * d + 2 * d * d + 2 * 2 * d * 4 * d + 2 * 2 * L * d)
    return L * per_tok + 2 * d * 512


def synthetic_batch(B, L, C, seed, dev):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, 224, 224, generator=g)
    text = torch.zeros(B, L, dtype=torch.int64)
    n = torch.randint(3, L, (B,), generator=g)
    for i in range(B):
        text[i, 0] = 49406
        text[i, 1:n[i]] = torch.randint(1, 49406, (int(n[i]) - 1,), generator=g)
        text[i, n[i]] = 49407
    label = (torch.rand(B, C, generator=g) < 0.15).float()
    return image.to(dev), text.to(dev), label.to(dev)


def cpu_baseline(L, bits, budget_s=10.0):
    """The reference's path on the host cores, encode+hash pairs/s, in the two forms SURVEY 8(d) asks for: the numpy restatement
    (oracle/clip_oracle.py) and the same modules on PyTorch CPU ops with all cores (oracle/torch_cpu.py = what upstream's CPU run
    executes).  `value` is the faster one."""
    import recipe
    import torch
    from oracle import clip_oracle as co
    from oracle.torch_cpu import TorchClip, linear_hash_codes
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    sd = recipe.clip_state_dict(recipe.CLIP_VITB32, 11)
    wi, bi = recipe.head_linear(512, bits, 3, "bi")
    wt, bt = recipe.head_linear(512, bits, 3, "bt")

    def timed(one, B, budget):
        t0 = time.perf_counter()
        one()                                           # warm-up, but it counts if it alone exhausts the budget
        first = time.perf_counter() - t0
        if first > budget:
            return B, first
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget:
            one()
            n += B
        return n, time.perf_counter() - t0

    B = 8
    img, txt = recipe.images(B, 224, 3), recipe.captions(B, L, 49408, 3)

    def one_numpy():
        hi = co.linear_hash(co.encode_image(sd, img), wi, bi)
        ht = co.linear_hash(co.encode_text(sd, txt), wt, bt)
        return co.sign_codes(hi), co.sign_codes(ht)
    n_np, dt_np = timed(one_numpy, B, budget_s * 0.5)
    Bt = 32                                                # configs[0]'s batch
    tc = TorchClip(sd)
    img_t, txt_t = torch.from_numpy(recipe.images(Bt, 224, 4)), torch.from_numpy(recipe.captions(Bt, L, 49408, 4))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 16)))       # a 1-GPU box of the pool owns a 16-core share: more threads only thrash

    def one_torch():
        return linear_hash_codes(tc.encode_image(img_t), wi, bi), linear_hash_codes(tc.encode_text(txt_t), wt, bt)
    n_t, dt_t = timed(one_torch, Bt, budget_s)
    v_np, v_t = n_np / dt_np, n_t / dt_t
    return {"value": round(max(v_np, v_t), 2), "unit": "pairs/s", "cores": int(torch.get_num_threads() if v_t >= v_np else cores), "kind": "port",
            "sample": f"PyTorch CPU ops ({torch.get_num_threads()} threads): {n_t} pairs in {dt_t:.1f} s (batches of {Bt}) = {v_t:.1f} pairs/s; "
                      f"numpy restatement: {n_np} pairs in {dt_np:.1f} s (batches of {B}) = {v_np:.1f} pairs/s; 224x224 + {L} tokens, ViT-B/32 fp32"}


def self_launch(n_gpus):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: start N ranks of this same command under
    torch.distributed.run (one process per GPU, RCCL) and pass rank 0's JSON line through.  This parent process never touches
    the GPU (nothing here initialises HIP); children are fresh processes, and a failed group is replaced by a fresh group once,
    on a new port, before giving up with a non-zero exit code."""
    import socket
    import subprocess
    last, first_rc = None, None
    for attempt in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CMH_BENCH_CHILD="1")
        last = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in last.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if last.returncode == 0 and lines:
            line = lines[-1]
            if attempt > 0:        # a failed first group must be visible in the line the driver parses, not only on stderr
                rec = json.loads(line)
                rec["relaunched"] = True
                rec["first_rc"] = first_rc
                line = json.dumps(rec)
            print(line, flush=True)
            return 0
        if first_rc is None:
            first_rc = last.returncode
        print(f"bench.py: {n_gpus}-rank group exited with code {last.returncode} (attempt {attempt + 1})", file=sys.stderr, flush=True)
    return last.returncode or 1


def flip_rates(clip, heads, image, text):
    """Sign-bit disagreement of the K-bit codes between the benchmarked arithmetic mode and the f32 parity mode (the reference's
    model.float() arithmetic) on the bench batch, per tower: the number behind "bit-exact sign()" for a low-precision
    mode.  Also the cosine of the 512-d features."""
    import cmh_native as N
    mode = clip.gemm_dtype
    out = {}
    with torch.no_grad():
        feats = {}
        for m in (mode, "f32"):
            clip.set_gemm_dtype(m)
            feats[m] = (clip.encode_image(image).float(), clip.encode_text(text).float())
        clip.set_gemm_dtype(mode)
        tot_flip = tot_bits = 0
        for side, head, k in (("image", heads[0], 0), ("text", heads[1], 1)):
            ca, cb = N.sign_codes(head(feats[mode][k])), N.sign_codes(head(feats["f32"][k]))
            flips = int((ca != cb).sum().item())
            tot_flip += flips
            tot_bits += ca.numel()
            cos = torch.nn.functional.cosine_similarity(feats[mode][k], feats["f32"][k], dim=1)
            out[side] = {"flip_rate": round(flips / ca.numel(), 6), "flipped_bits": flips, "bits": ca.numel(),
                         "feature_cosine_min": round(float(cos.min()), 6), "feature_cosine_mean": round(float(cos.mean()), 6)}
        out["both_towers"] = round(tot_flip / tot_bits, 6)
        out["samples"] = int(image.shape[0])
        out["what"] = f"{heads[0].fc.weight.shape[0]}-bit sign() codes of the bench batch, {mode} mode vs f32 mode, random-init weights"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of exactly --steps steps is run this many times; value = the MEDIAN region (value_runs lists min / median / max)")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (configs[1]: 256)")
    ap.add_argument("--seq-len", type=int, default=77)
    ap.add_argument("--bits", type=int, default=64)
    ap.add_argument("--classes", type=int, default=24)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-map-eval", action="store_true")
    ap.add_argument("--no-train-step", action="store_true")
    ap.add_argument("--no-input-pipeline", action="store_true")
    ap.add_argument("--no-dense-text", action="store_true", help="skip the extra timed pass with all text positions computed")
    ap.add_argument("--train-step", action="store_true",
                    help="also time the training step when --gpus > 1 (default: single-GPU runs only, so that the secondary "
                         "metric's gradient all-reduce can never stall the headline scaling line)")
    ap.add_argument("--towers", default="pair2", choices=["pair", "pair2", "streams", "pipelined", "serial"],
                    help="how a step runs the two towers: pair = in lock-step on one stream, layer i of both sharing its launches "
                         "(cmh_clip_encode_pair); pair2 (default) = the same, consecutive independent batches alternating between two HIP "
                         "streams - the product's encode loop (train/base.py::_code_loop); streams = one HIP stream per tower (rounds 1-3); "
                         "pipelined = streams without the per-step join of the side streams; serial = tower after tower on one stream")
    ap.add_argument("--no-overlap-towers", action="store_true", help="= --towers serial")
    ap.add_argument("--no-towers-ab", action="store_true", help="skip the timed legs of the other tower modes (streams, pair)")
    ap.add_argument("--no-precision-legs", action="store_true", help="skip flip_rate_vs_f32 and the timed f32-mode / fp8-mode legs")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the one-number-per-BASELINE-config legs (bench_configs.py)")
    ap.add_argument("--map-queries", type=int, default=5000)
    ap.add_argument("--map-db", type=int, default=15015)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:     # started bare: become the launcher of N ranks (before any GPU call)
        sys.exit(self_launch(a.gpus))

    import cmh_native as N
    import dist_utils as du
    from model.base.model import CLIP
    from model.modelbase import LinearHash
    from streams import overlapped
    from train.DSPH.loss import HyP

    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    rank, world, _ = du.init_from_env()
    dist_on = du.active()     # collectives run: more than one rank - or CMH_FORCE_DIST=1, which executes the N > 1 branch on RCCL in a group of one
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    B, L, K, C = a.batch, a.seq_len, a.bits, a.classes

    torch.manual_seed(1814)
    clip = CLIP(**VITB32).to(dev).float().set_gemm_dtype(a.dtype)
    clip.assume_frozen = True
    img_head, txt_head = LinearHash(512, K).to(dev).eval(), LinearHash(512, K).to(dev).eval()
    hyp = HyP(numclass=C, output_dim=K, hypseed=0, alpha=0.8).to(dev)
    image, text, label = synthetic_batch(B, L, C, 1814 + rank, dev)

    def tower(encode, head, x):
        h = head(encode(x))
        N.pack_codes(N.sign_codes(h), validate=False)
        return h

    towers = "serial" if a.no_overlap_towers else a.towers

    def finish(head, feat):
        h = head(feat)
        N.pack_codes(N.sign_codes(h), validate=False)
        return h

    n_pair_streams = max(1, int(os.environ.get("CMH_PAIR_STREAMS", "2")))      # (diagnostic: more alternating streams)
    pair_streams = [torch.cuda.Stream(device=dev) for _ in range(n_pair_streams)]
    import streams as _streams_mod
    _streams_mod._streams[torch.cuda.current_device()] = list(pair_streams)    # every mode of this process uses the SAME side streams: a
    # second set would share hardware queues with the first (the streams leg of towers_ab read 15 % slow that way)
    pair_turn = [0]

    def step(overlap=None, how=None):
        how = how or ("serial" if overlap is False else towers)
        with torch.no_grad():
            if how == "pair2":
                # the lock-step pair path, consecutive (independent) batches alternating between two streams: the grouped GEMMs of one
                # batch run beside the LayerNorm / attention launches and the launch gaps of the other
                s2 = pair_streams[pair_turn[0] % n_pair_streams]
                pair_turn[0] += 1
                cur = torch.cuda.current_stream(dev)
                with torch.cuda.stream(s2):
                    fi, ft = clip.encode_pair(image, text)
                    hi, ht = finish(img_head, fi), finish(txt_head, ft)
                cur.wait_stream(s2)          # the exchange step and the loss stay on the caller's stream
                hi.record_stream(cur)
                ht.record_stream(cur)
            elif how == "pair":       # both towers in lock-step: layer i of both is one grouped GEMM launch (csrc/encoders.hip)
                fi, ft = clip.encode_pair(image, text)
                hi, ht = finish(img_head, fi), finish(txt_head, ft)
            elif how in ("streams", "pipelined"):   # the two towers are independent until the loss: one HIP stream each (streams.py)
                hi, ht = overlapped(lambda: tower(clip.encode_image, img_head, image),
                                    lambda: tower(clip.encode_text, txt_head, text), inputs_ready=True if how == "pipelined" else None)
            else:
                hi = tower(clip.encode_image, img_head, image)
                ht = tower(clip.encode_text, txt_head, text)
            if dist_on:   # the path's one exchange step: fused all-gather of the per-rank code blocks
                fused, widths = du.fuse_columns(hi, ht, label)
                hi_g, ht_g, lab_g = du.split_columns(du.all_gather_rows(fused), widths)
            else:
                hi_g, ht_g, lab_g = hi, ht, label
            return hyp(hi_g, ht_g, lab_g)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if dist_on:   # the first use of a collective builds its rings: keep that out of the timed region whatever --warmup says
        du.all_gather_rows(torch.zeros(B, 2 * K + C, device=dev))
        barrier()
    step()          # set-up, not a step of the measurement: the first pass builds the bf16 weight copies and sizes the workspaces
    for _ in range(a.warmup):
        step()
    barrier()
    overlap = towers in ("streams", "pipelined", "pair2")      # per-launch events overlap only when kernels of two streams share the GPU
    reps = max(1, a.repeats)
    if not overlap:
        N.prof_gemm_begin(a.steps * reps * 128)

    def max_over_ranks(seconds):
        el = torch.tensor([seconds], dtype=torch.float64, device=dev)
        if dist_on:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        return float(el.item())

    # The timed region: EXACTLY --steps steps between two barrier + synchronize pairs, MAX over the ranks - run `reps` times back to
    # back (boxes of this pool differ by 10 % and the clock moves with the load: one 0.08 s sample cannot show a 1-3 % change).
    # `value` / `ms_per_step` are the MEDIAN region's; `value_runs` carries min / median / max of the same regions.
    region_s = []
    for _ in range(reps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = step()
        barrier()
        region_s.append(max_over_ranks(time.perf_counter() - t0))
    if overlap:
        # With the towers on two streams a GEMM's start-to-end time includes the time its workgroups wait for CUs held by
        # the other tower's kernel, so per-launch HIP events (and rocprofv3's kernel trace) no longer time the kernel.  The
        # roofline leg therefore re-runs the same K steps with the towers serialized, right after the timed region.
        N.prof_gemm_begin(a.steps * 128)
        for _ in range(a.steps):
            step(how="pair" if towers == "pair2" else "serial")
        barrier()
    prof_steps = a.steps if overlap else a.steps * reps
    all_ms, all_flops, all_launches = N.prof_gemm_end()
    by_kernel = N.prof_gemm_by_kernel()
    # the dominant kernel is gemm_wide_kernel (every GEMM of M >= 10 k rows); the M = 256 launches of the pooled-row tail and of the
    # final projections run on gemm_rows_kernel and are reported beside it, not averaged into it
    gemm_ms, gemm_flops, gemm_launches = by_kernel["gemm_wide_kernel"]
    srt = sorted(region_s)
    elapsed = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    assert torch.isfinite(loss).item(), "non-finite loss"

    pairs = a.steps * B * world
    value = pairs / elapsed
    # encode_text skips the padding after each caption's EOT (bit-identical pooled features): flops are counted on
    # the rows that were computed; `value_dense_text` below times the same steps with every one of the L positions computed
    rows_c, rows_d = clip.last_text_rows if clip.last_text_rows else (B * L, B * L)
    flops_pair = FLOP_IMG + text_flops(L) * rows_c / rows_d
    value_dense = None
    if clip.pack_text and not a.no_dense_text:
        clip.pack_text = False
        for _ in range(2):
            step()
        barrier()
        td0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        barrier()
        value_dense = pairs / max_over_ranks(time.perf_counter() - td0)
        clip.pack_text = True
    peak = PEAK_TFLOPS[a.dtype]
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0

    def gemm_algorithmic_bytes(text_rows):
        """operand + result bytes of the gemm_wide_kernel launches of one step, each tensor counted once (bf16 mode: bf16 operands
        and QKV / c_fc outputs, fp16 residual stream read + written by out_proj / c_proj, f32 output of conv1).  Per tower: 11 full
        blocks + the last block's QKV; the last block's other three GEMMs and the final projections have M = batch rows and run on
        gemm_rows_kernel."""
        e = 2 if a.dtype == "bf16" else 4
        xs = 2 if a.dtype == "bf16" else 4            # residual stream element
        total, n = 0, 0
        for M, d in ((B * 50, 768), (text_rows, 512)):
            shapes = ((3 * d, d, 0, e), (d, d, 1, xs), (4 * d, d, 0, e), (d, 4 * d, 1, xs))
            for layer in range(12):
                for i, (Nn, K, res, osz) in enumerate(shapes):
                    if layer == 11 and i > 0:
                        continue
                    total += M * K * e + Nn * K * e + M * Nn * osz + res * M * Nn * xs
                    n += 1
        total += B * 49 * 3072 * e + 768 * 3072 * e + B * 49 * 768 * 4                         # conv1 as a GEMM
        if towers in ("pair", "pair2"):      # layer i of both towers is ONE launch: 45 grouped launches + conv1
            n //= 2
        return total, n + 1
    # HBM-side bytes per GEMM launch cannot be counted from inside this process: they come from the two rocprofv3 --pmc
    # passes of this same command (tools/pmc_bench_traffic.sh), committed under profiles/; null if absent / other dtype.
    traffic, traffic_src = None, None
    import glob
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic.json")))      # the latest round's passes
    if a.dtype == "bf16" and tfiles:
        with open(tfiles[-1]) as fh:
            traffic = json.load(fh)["traffic_bytes_per_launch"]
        traffic_src = f"profiles/{os.path.basename(tfiles[-1])} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, (2*FETCH+WRITE)*1024)"
    out = {
        "metric": "image+text pairs/s encoded+hashed per GPU; mAP@K eval wallclock (64-bit)",
        "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "value_runs": {"pairs_per_s": [round(pairs / x, 2) for x in (srt[-1], elapsed, srt[0])], "ms_per_step": [round(x / a.steps * 1e3, 4) for x in (srt[0], elapsed, srt[-1])],
                       "regions": reps, "steps_per_region": a.steps,
                       "what": "[min, median, max] over back-to-back timed regions of exactly --steps steps each; value / ms_per_step = the median region"},
        "collectives": (None if not dist_on else torch.distributed.get_backend() + (" (forced group of one)" if world == 1 else "")),
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "residual_stream": ("fp16" if a.dtype == "bf16" and os.environ.get("CMH_RESID_F16", "1") != "0" else "f32"),
        "arithmetic": ("bf16 MFMA operands, f32 accumulate, f32 LayerNorm statistics / softmax, fp16 residual stream" if a.dtype == "bf16"
                       else "f32 operands on the f32 MFMA (exact fma chain), f32 everywhere: the reference's model.float() arithmetic"),
        "config": {"workload": "configs[1]: DSPH flickr25k output-dim 64, ViT-B/32, batch 256/GPU, 224x224 + "
                               f"{L}-token captions: encode_image+encode_text -> LinearHash -> sign -> pack -> "
                               "[all-gather] -> HyP loss fwd", "per_gpu_batch": B, "global_batch": B * world,
                   "seq_len": L, "bits": K, "weights": "random-init ViT-B/32", "parallelism": f"batch-shard x{world}",
                   "towers": towers,
                   "streams": {"pair": "one stream, the towers in lock-step: layer i of both shares its GEMM launches (grouped)",
                               "streams": "image and text tower on one HIP stream each",
                               "pair2": "the lock-step pair path; consecutive (independent) batches alternate between two HIP streams",
                               "pipelined": "image and text tower on one HIP stream each; consecutive (independent) batches overlap: a tower's stream "
                                            "starts batch n + 1 as soon as it has finished batch n", "serial": "single stream, tower after tower"}[towers]},
        "per_gpu_value": round(value / world, 2),
        "text_rows": {"computed": rows_c, "dense": rows_d,
                      "note": "caption tokens after the EOT cannot reach the pooled feature under the causal mask; they are not "
                              "computed (synthetic captions: EOT uniform in positions 3..75)"},
        "value_dense_text": None if value_dense is None else round(value_dense, 2),
        "end_to_end_tflops_per_gpu": round(value / world * flops_pair / 1e12, 2),
        "roofline": {"bound": "mfma", "kernel": "cmh::gemm_wide_kernel<%s, 1, *>%s" % ("0" if a.dtype == "f32" else "1", " (grouped launches: layer i of both "
                     "towers per launch)" if towers in ("pair", "pair2") else ""),
                     "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                     "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": round(gemm_algorithmic_bytes(rows_c)[0] / gemm_algorithmic_bytes(rows_c)[1]),
                     "launches": int(gemm_launches),
                     "avg_launch_us": round(gemm_ms * 1e3 / max(gemm_launches, 1), 2),
                     "timing": "HIP events stamped by the dispatch itself (hipExtLaunchKernelGGL start/stop events on the launch "
                               "stream): the kernel's own begin-to-end time, as rocprofv3's kernel trace reports it",
                     "other_gemm_kernels": {k: {"launches": int(v[2]), "ms_per_step": round(v[0] / prof_steps, 4),
                                                "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else 0.0}
                                            for k, v in by_kernel.items() if k != "gemm_wide_kernel" and v[2] > 0},
                     "all_gemm_launches": {"launches": int(all_launches), "ms_per_step": round(all_ms / prof_steps, 4),
                                           "tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0},
                     "measured_in": ("the timed region" if not overlap else
                                     "a second pass of the same K steps on ONE stream (the lock-step pair path, batch after batch): in the "
                                     "timed region the launches of two batches overlap, so per-launch events would not time the kernel; "
                                     "value/ms_per_step are from the overlapped region (towers_ab.pair holds a roofline measured in ITS "
                                     "timed region)" if towers == "pair2" else
                                     "a second pass of the same K steps with the two towers serialized (per-launch events "
                                     "overlap when the towers share the GPU); value/ms_per_step are from the overlapped region"),
                     "gemm_ms_per_step_serialized": round(gemm_ms / prof_steps, 4)},
    }

    if world == 1 and not a.no_towers_ab:
        # The other ways to run a step's two towers, timed on this very box (one region of --steps steps each), and - in the single-stream
        # lock-step mode, where launches do not overlap - the GEMM roofline measured IN that timed region by the dispatches' own events
        ab = {}
        for how in ("streams", "pair", "pair2"):
            if how == towers:
                continue
            try:
                for _ in range(3):
                    step(how=how)
                barrier()
                if how == "pair":
                    N.prof_gemm_begin(a.steps * 128)
                tb0 = time.perf_counter()
                for _ in range(a.steps):
                    step(how=how)
                barrier()
                tb_ms = (time.perf_counter() - tb0) / a.steps * 1e3
                ab[how] = {"pairs_per_s": round(B / tb_ms * 1e3, 2), "ms_per_step": round(tb_ms, 4), "steps": a.steps,
                           "vs_headline": round(out["ms_per_step"] / tb_ms, 4)}
                if how == "pair":
                    p_ms, p_fl, p_n = N.prof_gemm_end()
                    pk = N.prof_gemm_by_kernel()["gemm_wide_kernel"]
                    ab[how]["roofline"] = {"bound": "mfma", "achieved": round(pk[1] / (pk[0] * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                           "frac": round(pk[1] / (pk[0] * 1e-3) / 1e12 / peak, 4), "launches": int(pk[2]),
                                           "avg_launch_us": round(pk[0] * 1e3 / max(pk[2], 1), 2), "measured_in": "the timed region",
                                           "all_gemm_launches": {"launches": int(p_n), "ms_per_step": round(p_ms / a.steps, 4),
                                                                 "tflops": round(p_fl / (p_ms * 1e-3) / 1e12, 2) if p_ms > 0 else 0.0}}
            except Exception as exc:
                ab[how] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        ab["what"] = ("streams: one HIP stream per tower (the default of rounds 1-3); pair: both towers in lock-step on ONE stream, layer i of both "
                      "as one grouped launch (GEMM, LayerNorm, attention) - launches do not overlap, so its roofline is measured in the timed "
                      "region; pair2: pair with consecutive batches alternating between two streams")
        out["towers_ab"] = ab

    if world == 1 and a.dtype != "f32" and not a.no_precision_legs:      # (step() holds a collective when world > 1: single-GPU runs only)
        try:
            # the precision story of the benchmarked mode: sign flips against the f32 parity mode on this very batch, and the same
            # step timed in f32 mode (towers serialized so that the per-launch events time the kernels)
            out["flip_rate_vs_f32"] = flip_rates(clip, (img_head, txt_head), image, text)
            clip.set_gemm_dtype("f32")
            nf = max(2, min(a.steps, 10))
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for _ in range(nf):
                step()
            torch.cuda.synchronize()
            f32_ms = (time.perf_counter() - tf0) / nf * 1e3
            N.prof_gemm_begin(nf * 128)
            for _ in range(nf):
                step(how="pair" if towers in ("pair", "pair2") else "serial")
            torch.cuda.synchronize()
            N.prof_gemm_end()
            g_ms, g_fl, g_n = N.prof_gemm_by_kernel()["gemm_wide_kernel"]
            f32_tf = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
            out["f32_mode"] = {"pairs_per_s": round(B / f32_ms * 1e3, 2), "ms_per_step": round(f32_ms, 3), "steps": nf,
                               "roofline": {"bound": "mfma", "achieved": round(f32_tf, 2), "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                                            "frac": round(f32_tf / PEAK_TFLOPS["f32"], 4), "launches": int(g_n)},
                               "what": "the same step with set_gemm_dtype('f32'): the parity mode (v_mfma_f32_16x16x4_f32, f32 residual stream)"}
            clip.set_gemm_dtype(a.dtype)
            step()
        except Exception as exc:
            out["f32_mode"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            clip.set_gemm_dtype(a.dtype)
        try:
            # BASELINE configs[4]'s encoder arithmetic on the same workload: the blocks' four GEMMs on e4m3 operands (fp8 MFMA),
            # scales calibrated on ANOTHER seeded batch; timed like the headline (towers overlapped), roofline leg serialized
            clip.set_gemm_dtype("fp8")
            cal_image, cal_text, _ = synthetic_batch(B, L, C, 99991 + rank, dev)     # NOT the measured batch
            with torch.no_grad():
                clip.calibrate_fp8(image=cal_image, text=cal_text)
            del cal_image, cal_text
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            fp8_ms = (time.perf_counter() - t8) / a.steps * 1e3
            N.prof_gemm_begin(a.steps * 128)
            for _ in range(a.steps):
                step(how="pair" if towers in ("pair", "pair2") else "serial")
            torch.cuda.synchronize()
            N.prof_gemm_end()
            by8 = N.prof_gemm_by_kernel()
            g_ms, g_fl, g_n = by8["gemm_wide_kernel"]
            fp8_tf = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
            fp8_other = {k: {"launches": int(n), "ms_per_step": round(ms / a.steps, 4)} for k, (ms, fl, n) in by8.items()
                         if k != "gemm_wide_kernel" and n > 0}
            out["fp8_mode"] = {"pairs_per_s": round(B / fp8_ms * 1e3, 2), "ms_per_step": round(fp8_ms, 4), "steps": a.steps,
                               "speedup_vs_headline": round(out["ms_per_step"] / fp8_ms, 3),
                               "roofline": {"bound": "mfma", "kernel": "cmh::gemm_wide_kernel<2, *>", "achieved": round(fp8_tf, 2),
                                            "peak": PEAK_TFLOPS["fp8"], "unit": "TFLOP/s", "frac": round(fp8_tf / PEAK_TFLOPS["fp8"], 4),
                                            "launches": int(g_n), "gemm_ms_per_step_serialized": round(g_ms / a.steps, 4),
                                            "other_gemm_kernels": fp8_other,
                                            "note": "the gemm_wide_kernel launches of the step against the dense fp8 peak (the few-row "
                                                    "launches of the pooled tail run on gemm_rows_kernel, listed beside it); conv1 "
                                                    "(2 % of the FLOPs) stays bf16"},
                               "flip_rate_vs_f32": flip_rates(clip, (img_head, txt_head), image, text),
                               "what": "set_gemm_dtype('fp8'): QKV / out_proj / c_fc / c_proj on v_mfma_scale_f32_16x16x128_f8f6f4 "
                                       "(e4m3 x e4m3, f32 accumulate), per-channel weight scales, per-tensor activation scales "
                                       "calibrated on a different seeded batch of the same size; fp16 residual stream, bf16 attention"}
            clip.set_gemm_dtype(a.dtype)
            step()
        except Exception as exc:
            out["fp8_mode"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            clip.set_gemm_dtype(a.dtype)

    if not a.no_map_eval:
        # secondary metric: 4 x calc_map_k at MIRFlickr scale, 64-bit, codes resident; queries sharded over ranks
        Q, Nn = a.map_queries, a.map_db
        g = torch.Generator().manual_seed(1234)
        rL = (torch.rand(Nn, C, generator=g) < 0.15).float()
        qL = (torch.rand(Q, C, generator=g) < 0.15).float()
        W = torch.randn(C, K, generator=g)
        mk = lambda lab: torch.sign(lab @ W + 0.5 * torch.randn(lab.shape[0], K, generator=g) + 1e-3).to(dev)
        r_img, r_txt, q_img, q_txt = mk(rL), mk(rL), mk(qL), mk(qL)
        lo, hi_ = du.shard_range(Q, rank, world)
        rLp, qLp = N.pack_labels(rL.to(dev)), N.pack_labels(qL[lo:hi_].to(dev))
        planes = {k: N.pack_codes(v) for k, v in dict(r_img=r_img, r_txt=r_txt, q_img=q_img[lo:hi_], q_txt=q_txt[lo:hi_]).items()}

        def four(tie=N.TIE_REFERENCE):
            res = []
            for qk, rk in (("q_img", "r_txt"), ("q_txt", "r_img"), ("q_img", "r_img"), ("q_txt", "r_txt")):
                mp, ap_l, _ = N.hamming_map(planes[qk], qLp, planes[rk], rLp, K, C, tie_order=tie)
                # one rank: the kernel's own mean; several: the same sequential f32 sum over the gathered per-query APs (cmh_map_mean)
                res.append(du.mean_in_query_order(du.gather_query_sharded_ap(ap_l, Q)) if dist_on else mp)
            return res

        def timed(tie):
            four(tie)
            barrier()
            t0 = time.perf_counter()
            maps = four(tie)
            barrier()
            return (time.perf_counter() - t0) * 1e3, maps
        ms, maps = timed(N.TIE_REFERENCE)
        ms_st, maps_st = timed(N.TIE_STABLE)
        alg_bytes = 4 * ((Q + Nn) * K // 8 + (Q + Nn) * ((C + 7) // 8) + 4 * Q)      # packed inputs once + per-query AP
        out["map_eval"] = {"ms": round(ms, 3), "directions": 4, "Q": Q, "N": Nn, "bits": K,
                           "code_pairs_per_s": round(4.0 * Q * Nn / (ms * 1e-3), 1),
                           "algorithmic_GBps": round(alg_bytes / (ms * 1e-3) / 1e9, 3),
                           "bound": "ranking (emulated introsort per query), not HBM: the packed inputs are %.2f MB" % (alg_bytes / 1e6),
                           "tie_order": "reference (libstdc++ introsort)", "mAP_i2t": round(float(maps[0]), 6),
                           "stable_tie_order": {"ms": round(ms_st, 3), "mAP_i2t": round(float(maps_st[0]), 6),
                                                "note": "CMH_TIE_STABLE (ties by index): not the reference's ranking"}}

    if rank == 0:
        # SURVEY 8(d), N x N loss kernels: time per step and share of the step (HyP on the rank's batch; tools/loss_scale_bench.py
        # has the global-batch sizes of an N-GPU step)
        with torch.no_grad():
            hi_l, ht_l = img_head(torch.randn(B, 512, device=dev)), txt_head(torch.randn(B, 512, device=dev))
            for _ in range(3):
                hyp(hi_l, ht_l, label)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hyp(hi_l, ht_l, label)
            e1.record()
            torch.cuda.synchronize()
            lus = e0.elapsed_time(e1) / 20 * 1e3
        out["loss_fwd"] = {"us": round(lus, 1), "share_of_step": round(lus * 1e-3 / out["ms_per_step"], 4), "what": f"HyP forward, batch {B}, {K} bits, {C} classes"}

    if not a.no_train_step and (world == 1 or a.train_step):
        try:
            # secondary metric: the reference's actual inner loop (train/DSPH/hash_train.py:49-73) - tape-keeping forward of both
            # towers, heads, HyP loss, backward through everything, fused BertAdam + SGD on the proxies; same batch, weights not frozen
            from model.base.optimization import BertAdam
            clip.assume_frozen = False
            img_head.train(); txt_head.train()
            params = [p for n, p in clip.named_parameters() if n != "logit_scale"]
            opt = BertAdam([{"params": params, "lr": 1e-5}, {"params": list(img_head.parameters()) + list(txt_head.parameters())}],
                           lr=1e-3, warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=1000, weight_decay=0.2,
                           max_grad_norm=1.0)
            sgd = torch.optim.SGD(hyp.parameters(), lr=0.02, momentum=0.9, weight_decay=0.0005)

            # gradient means over the ranks (dist_utils.GradSync): the towers' gradients as in-place buckets of their flat buffer,
            # queued from inside backward part by part; the heads' through the hook route
            vis_ids = {id(p) for p in clip.visual.parameters()}
            sync = du.GradSync([[p for p in params if id(p) not in vis_ids], list(clip.visual.parameters()),
                                list(img_head.parameters()) + list(txt_head.parameters()) + list(hyp.parameters())])

            def train_step():
                fi, ft = overlapped(lambda: clip.encode_image(image), lambda: clip.encode_text(text))
                hi, ht = img_head(fi), txt_head(ft)
                # the exchange step exactly as the trainers run it (train/DSPH/hash_train.py:_step -> TrainBase.loss_inputs): one
                # fused, differentiable all-gather; the loss sees the global batch, gradients come back through the local rows
                hi_g, ht_g, lab_g = du.gather_loss_inputs(hi, ht, label)
                loss = hyp(hi_g, ht_g, lab_g)
                opt.zero_grad(); sgd.zero_grad()
                loss.backward()
                sync.finish()                            # no-op on one GPU
                opt.step(); sgd.step()
                return loss
            for _ in range(2):
                train_step()
            barrier()
            t0 = time.perf_counter()
            nts = max(20, a.steps)
            for _ in range(nts):
                tl = train_step()
            barrier()
            tms = max_over_ranks(time.perf_counter() - t0) / nts * 1e3
            out["train_step"] = {"ms": round(tms, 3), "pairs_per_s": round(B * world / tms * 1e3, 1), "steps": nts,
                                 "what": "DSPH step: forward with tape + HyP loss + backward (heads, both towers) + fused BertAdam",
                                 "loss": round(float(tl.detach()), 5)}
            clip.assume_frozen = True
            sync.remove()
        except Exception as exc:      # the secondary metric must never take the headline line down
            out["train_step"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            clip.assume_frozen = True

    if rank == 0 and not a.no_input_pipeline:
        try:
            # secondary metric: the input pipeline that feeds the step — image transform of a raw batch on the GPU
            # (500x375 / 375x500 decoded images -> 224x224 normalised floats, bit-identical to Pillow + torchvision) and the
            # native batch tokenizer.  The tokenizer needs the CLIP merges file (data, not shipped): reported when it is found.
            import numpy as np
            from dataset.gpu_transform import RaggedImages, preprocess
            rng = np.random.default_rng(0)
            raw = RaggedImages.from_arrays([rng.integers(0, 256, ((375, 500) if i % 3 else (500, 375)) + (3,), dtype=np.uint8)
                                            for i in range(B)])
            on_dev = raw.to(dev)
            for _ in range(3):
                preprocess(on_dev, 224, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                preprocess(on_dev, 224, True)
            torch.cuda.synchronize()
            ms_res = (time.perf_counter() - t0) / 10 * 1e3
            t0 = time.perf_counter()
            for _ in range(5):
                preprocess(raw.to(dev), 224, True)
            torch.cuda.synchronize()
            ms_h2d = (time.perf_counter() - t0) / 5 * 1e3
            out["input_pipeline"] = {"image_transform_ms": round(ms_res, 3), "images_per_s": round(B / ms_res * 1e3, 1),
                                     "with_pinned_h2d_ms": round(ms_h2d, 3), "images_per_s_with_h2d": round(B / ms_h2d * 1e3, 1),
                                     "GBps_in_plus_out": round((raw.pixels.numel() + B * 3 * 224 * 224 * 4) / ms_res / 1e6, 1),
                                     "what": f"{B} decoded RGB images (500x375 mix) -> Resize(224, BICUBIC) + CenterCrop + ToTensor + Normalize on the GPU"}
            from dataset.gpu_transform import normalize_u8
            cache = torch.randint(0, 256, (4 * B, 224, 224, 3), dtype=torch.uint8, device=dev)     # device-resident resized images
            pick = torch.randperm(4 * B, device=dev)[:B]
            for _ in range(3):
                normalize_u8(cache, pick)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                normalize_u8(cache, pick)
            torch.cuda.synchronize()
            ms_c = (time.perf_counter() - t0) / 10 * 1e3
            out["input_pipeline"]["cached_epoch_batch_ms"] = round(ms_c, 3)
            out["input_pipeline"]["cached_images_per_s"] = round(B / ms_c * 1e3, 1)
            # the native batch tokenizer (cmh_bpe_encode_captions).  Its merges table is DATA the user supplies (the CLIP vocabulary is
            # not shipped); when it is absent the same code path is timed on the repository's own miniature merges file - fewer merges
            # per word, so it reads somewhat faster than the real vocabulary would; the field says which one was used
            from model.base.simple_tokenizer import SimpleTokenizer, default_bpe
            vocab = default_bpe()
            which = "the CLIP merges file (bpe_simple_vocab_16e6)"
            if not os.path.exists(vocab):
                vocab = os.path.join(ROOT, "tests", "golden", "mini_bpe_merges.txt")
                which = "tests/golden/mini_bpe_merges.txt (270 merges; the CLIP merges file is not shipped)"
            tok = SimpleTokenizer(vocab)
            words = "a man riding a wave on top of a surfboard while two dogs play in the snow near the old red barn".split()
            caps = [" ".join(rng.choice(words, size=int(rng.integers(8, 25)))) for _ in range(20000)]
            tok.encode_captions(caps[:2000], 32)
            t0 = time.perf_counter()
            tok.encode_captions(caps, 32)
            out["input_pipeline"]["captions_per_s"] = round(len(caps) / (time.perf_counter() - t0), 1)
            out["input_pipeline"]["captions_vocabulary"] = which
        except Exception as exc:
            out["input_pipeline"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    if rank == 0 and world == 1 and not a.no_config_legs:
        # one timed number for each of the other BASELINE.json configs (the headline above is configs[1]); each leg builds its own
        # model, so the headline's is released first
        import gc
        import bench_configs
        del clip, img_head, txt_head, hyp
        gc.collect()
        torch.cuda.empty_cache()
        for name, leg in bench_configs.LEGS:
            try:
                t_leg = time.perf_counter()
                if name == "dchmt_epoch":
                    out[name] = leg(dev, cpu_sample=not a.no_cpu_baseline)
                elif name == "code_loop":
                    out[name] = leg(dev, headline_pairs_per_s=out["value"], L=L, bits=K, batch=B)
                else:
                    out[name] = leg(dev)
                out[name]["leg_wallclock_s"] = round(time.perf_counter() - t_leg, 1)
            except Exception as exc:
                out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            gc.collect()
            torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(L, K)
        out["cpu_baseline"] = cb
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
