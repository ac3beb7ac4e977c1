#!/usr/bin/env python3
"""Register counts, scratch and an instruction hash of the GEMM kernels the default paths launch, from hipcc's own assembly.

    python3 tools/isa_pin.py record   # rewrite tests/golden/isa_pins.json from the current sources
    python3 tools/isa_pin.py check    # compare (exit 1 on any difference; `-v` prints the per-kernel table)

Why: the 160-row instantiations of gemm_wide_kernel sit at 256 VGPRs with a few bytes of scratch OUTSIDE the K loop; round 3 found
that an unrelated source edit (even one behind `if constexpr (false)`) can move the allocator and cost the headline 0.6 %.  This
record turns such a perturbation into a red CPU test (tests/test_isa_pins.py) instead of a mystery on the next box.  A deliberate
kernel change re-records the file in the same commit - together with a bench A/B (tools/lib_ab.sh).

What is hashed: the instruction lines of each kernel body as hipcc prints them for gfx950 (-S --cuda-device-only), comments
stripped, basic-block labels renumbered in order of appearance (their numbers depend on the position of the function in the TU)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "clip-based-cross-modal-hashing_amd", "csrc")
PINS = os.path.join(ROOT, "tests", "golden", "isa_pins.json")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=on", "-S", "--cuda-device-only",
         "-I", os.path.join(ROOT, "include")]
# (file, regex on the demangled kernel name): the kernels of the default encode / training step
WATCH = [("gemm_wide.hip", r"gemm_wide_kernel<|gemm_wide_tn_multi_kernel"), ("gemm_rows.hip", r"gemm_rows_kernel<")]


def assemble(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, os.path.join(CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels_of(asm):
    """{mangled name: {vgpr, sgpr, scratch, lds, instructions, sha256}}"""
    lines = asm.split("\n")
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        body = m.group(2)

        def field(name):
            mm = re.search(r"\.amdhsa_%s (\d+)" % name, body)
            return int(mm.group(1)) if mm else None
        meta[m.group(1)] = {"vgpr": field("next_free_vgpr"), "sgpr": field("next_free_sgpr"),
                            "scratch": field("private_segment_fixed_size"), "lds": field("group_segment_fixed_size")}
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if m and m.group(1) in meta:
            name = m.group(1)
            labels, body = {}, []
            i += 1
            while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
                ln = lines[i].split(";")[0].rstrip()
                i += 1
                t = ln.strip()
                if not t or (t.startswith(".") and not re.match(r"^\.LBB\d+_\d+:$", t)):   # directives (.amdhsa_*, .section ...) are not code
                    continue
                body.append(t)
            text = "\n".join(body)
            for lab in re.findall(r"\.LBB\d+_\d+", text):
                labels.setdefault(lab, ".L%d" % len(labels))
            text = re.sub(r"\.LBB\d+_\d+", lambda mm: labels[mm.group(0)], text)
            n_inst = sum(1 for b in body if not b.endswith(":"))
            out[name] = dict(meta[name], instructions=n_inst, sha256=hashlib.sha256(text.encode()).hexdigest()[:16])
        i += 1
    return out


def demangle(names):
    res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, res))


def short_names(ks):
    dm = demangle(list(ks))
    return {re.sub(r"\(.*", "", dm[k]).replace("void cmh::", ""): v for k, v in ks.items()}


def current():
    rec = {}
    for src, pat in WATCH:
        for short, v in short_names(kernels_of(assemble(src))).items():
            if re.search(pat, short):
                rec[short] = v
    return dict(sorted(rec.items()))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    cur = current()
    if mode == "record":
        with open(PINS, "w") as f:
            json.dump({"hipcc": subprocess.check_output(["/opt/rocm/bin/hipcc", "--version"], text=True).split("\n")[0],
                       "kernels": cur}, f, indent=1)
            f.write("\n")
        print(f"recorded {len(cur)} kernels -> {PINS}")
        return 0
    want = json.load(open(PINS))["kernels"]
    bad = 0
    for k in sorted(set(cur) | set(want)):
        a, b = want.get(k), cur.get(k)
        if a != b:
            bad += 1
            print(f"DIFF {k}\n   pinned  {a}\n   current {b}")
        elif "-v" in sys.argv:
            print(f"ok   {k}  {b}")
    print(f"{len(cur)} kernels, {bad} differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
