"""Per-kernel table of a MITH forward from a rocprofv3 kernel_stats.csv (bash tools/leg_trace.sh mith_step <tag> with CMH_LEG_FWD_ONLY=1
CMH_LEG_TOKENS=77: 7 forward passes), grouped by the part of model/MITH.py the kernel belongs to:
   python tools/mith_kernel_table.py <csv> <forward passes>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
GROUPS = [  # (label, substrings of the kernel name) - first match wins
    ("trunk + HashingModel GEMMs (gemm_wide / gemm_rows / small_linear)", ("gemm_wide_kernel", "gemm_rows_kernel", "small_linear", "gemm_kernel")),
    ("trunk attention (12 + 12 layers) + concept transformer attention", ("attention_mfma_kernel", "attention_kernel")),
    ("LayerNorm (trunk ln_1 / ln_2 / ln_post / ln_final on every token; ResidualMLPs; concept transformer)", ("layernorm",)),
    ("LocalizedTokenAggregation: top-k + softmax + weighted merge (lta_kernel)", ("lta_kernel",)),
    ("BitwiseHashing (bitwise_hash) / l2 normalise / positional add", ("bitwise_hash", "l2_normalize", "add_positional")),
    ("casts to bf16 (GEMM operands the epilogues do not emit)", ("cast_bf16",)),
    ("patchify / embeddings / text pack plan", ("patchify", "embed", "assemble", "pack_plan", "gather_rows", "scatter_rows")),
    ("sign codes / packing", ("sign_codes", "pack_codes")),
    ("D2D copies and fills (torch, runtime)", ("copyBuffer", "fillBuffer", "FillFunctor", "elementwise", "CatArray", "direct_copy")),
]
acc = {g[0]: [0.0, 0] for g in GROUPS}
other = [0.0, 0, []]
for r in rows:
    for label, keys in GROUPS:
        if any(k in r["Name"] for k in keys):
            acc[label][0] += float(r["TotalDurationNs"]); acc[label][1] += int(r["Calls"])
            break
    else:
        other[0] += float(r["TotalDurationNs"]); other[1] += int(r["Calls"]); other[2].append(r["Name"][:40])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'part':110s} {'launches/fwd':>12s} {'ms/fwd':>8s} {'share':>6s}")
for label, (ns, calls) in acc.items():
    print(f"{label:110s} {calls / n:12.1f} {ns / n / 1e6:8.3f} {100 * ns / tot:5.1f}%")
print(f"{'other: ' + ', '.join(sorted(set(other[2])))[:100]:110s} {other[1] / n:12.1f} {other[0] / n / 1e6:8.3f} {100 * other[0] / tot:5.1f}%")
print(f"sum of kernel durations: {tot / n / 1e6:.3f} ms per forward (two streams overlap them: the leg's wall time is less)")
