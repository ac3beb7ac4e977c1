import os, sys, json, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.argv = ["bench.py"]
import torch, bench, bench_configs, recipe
from types import SimpleNamespace
import cmh_native as N
sys.path.insert(0, os.path.join(R, "tests"))
import mithutil as mu
from model.MITH import HashingModel, build_model
from streams import overlapped
dev = torch.device("cuda:0")
torch.manual_seed(0)
clip = build_model(bench_configs._vitb32_state(1)).to(dev).float().set_gemm_dtype("bf16")
clip.padded_tokens_unused = True      # as model/MITH.py::MITH sets it (HashingModel is the only reader of the text tokens)
hm = HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=64, **mu.ARGS)).to(dev).eval().set_gemm_dtype("bf16")
B = 256
img = torch.randn(B, 3, 224, 224, device=dev)
txt = torch.from_numpy(recipe.captions(B, 77, 49408, 1)).to(dev); kpm = txt == 0
def fwd():
    with torch.no_grad():
        (seq_i, _, cls_i), (seq_t, _, nk, eos) = overlapped(lambda: clip.encode_image(img), lambda: clip.encode_text(txt, kpm))
        return hm(seq_i, seq_t, cls_i, eos, nk)
for _ in range(3): fwd()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    fwd(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::fill_", "aten::zero_", "aten::zeros", "aten::empty_like"):
        st = [s for s in (e.stack or []) if "clip-based" in s or "cmh_native" in s or "mith" in s.lower()]
        cnt[(e.name, str(e.input_shapes)[:80], (st[0] if st else "")[-90:])] += 1
for k, v in cnt.most_common(40): print(v, k)
mem = [e for e in prof.events() if "Memcpy" in e.name or "copyBuffer" in e.name]
print("memcpy-like device events:", len(mem), collections.Counter(e.name for e in mem))
