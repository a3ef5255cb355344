#!/usr/bin/env python3
"""Static instruction mix of the hot loop of one raytrace kernel variant (gfx950 ISA from hipcc -S).

    python tools/isa_mix.py ["<256, false, false, false, 256, false, false, true, 2, false>"] [--dump loop.s]

Compiles pyc2ray_amd/csrc/raytrace.hip to assembly with the Makefile's flags, takes the named instantiation of
raytrace_octant_kernel, finds its LARGEST loop (the backward branch spanning the most instructions: the sweep over the steps
of a unit) and prints how many instructions of each class one trip through it issues per wave -- counting every basic block
once, so the figure is an upper bound where the loop body branches (the rate section is skipped by waves without work).
The dynamic count to hold beside it is SQ_INSTS_VALU per wave-step (profiles/*_pmc_*.txt).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pyc2ray_amd", "csrc", "raytrace.hip")
args = [a for a in sys.argv[1:]]
dump = None
if "--dump" in args:
    i = args.index("--dump")
    dump = args[i + 1]
    del args[i:i + 2]
want = args[0] if args else "<256, false, false, false, 256, false, false, true, 2, false>"
extra = os.environ.get("EXTRA", "").split()

with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "rt.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", *extra,
                    "-S", "--cuda-device-only", "-o", out, SRC], check=True, capture_output=True)
    text = open(out).read().splitlines()

# function bodies: "<mangled>:" ... ".Lfunc_end"
starts = [(n, l[:-1]) for n, l in enumerate(text) if re.match(r"^_Z\w+:$", l.split(";")[0].strip()) for l in [l.split(";")[0].strip()]]
names = subprocess.run(["c++filt"], input="\n".join(s for _, s in starts), capture_output=True, text=True).stdout.splitlines()
body = None
for (n, s), d in zip(starts, names):
    if "raytrace_octant_kernel" in d and want in d:
        end = next(m for m in range(n, len(text)) if text[m].startswith(".Lfunc_end"))
        body, title = text[n + 1:end], d
        break
if body is None:
    sys.exit("no instantiation matching " + want)

ins = []            # (label or None, opcode, operands)
labels = {}
for l in body:
    l = l.split(";")[0].rstrip()
    if not l.strip():
        continue
    m = re.match(r"^(\.L\w+):", l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if l.startswith("\t.") or l.startswith(" ."):
        continue
    parts = l.strip().split(None, 1)
    ins.append((parts[0], parts[1] if len(parts) > 1 else ""))

# the largest backward branch
best = None
for n, (op, rest) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        tgt = rest.strip()
        if tgt in labels and labels[tgt] <= n:
            span = n - labels[tgt]
            if best is None or span > best[1] - best[0]:
                best = (labels[tgt], n)
lo, hi = best
loop = ins[lo:hi + 1]
if dump:
    open(dump, "w").write("\n".join(f"{o}\t{r}" for o, r in loop) + "\n")


def klass(op):
    if op.startswith("v_"):
        if re.match(r"v_(fma|mul|add|fmac|max|min|ldexp|frexp_mant|fract|floor|trunc|rndne|div_fixup|div_fmas|div_scale)_f64", op) or op == "v_cvt_f64_i32" \
                or op == "v_cvt_f64_u32" or op == "v_cvt_i32_f64" or op == "v_cvt_u32_f64" or op.startswith("v_frexp_exp_i32_f64"):
            return "VALU f64 arithmetic"
        if re.match(r"v_(rcp|rsq|sqrt|exp|log)_f64", op):
            return "VALU f64 rcp/rsq/sqrt (quarter rate)"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "VALU compares"
        if op.startswith("v_cndmask"):
            return "VALU selects"
        if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") \
                or op.startswith("v_writelane") or op.startswith("v_swap"):
            return "VALU moves"
        if re.search(r"_f32|_f16", op):
            return "VALU f32"
        return "VALU integer / bit"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("buffer_atomic") or op.startswith("global_atomic") or op.startswith("flat_atomic"):
        return "memory atomics"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "memory loads/stores"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branches"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "scalar loads"
    if op.startswith("s_nop") or op.startswith("s_sleep"):
        return "s_nop"
    return "SALU"


mix = collections.Counter(klass(o) for o, _ in loop)
ops = collections.Counter(o for o, _ in loop)
print(title)
print(f"largest loop: {len(loop)} instructions (of {len(ins)} in the kernel)")
valu = sum(c for k, c in mix.items() if k.startswith("VALU"))
for k, c in sorted(mix.items(), key=lambda kv: -kv[1]):
    print(f"  {k:40s} {c:5d}")
print(f"  {'VALU in total':40s} {valu:5d}")
print("most frequent opcodes: " + ", ".join(f"{o} x{c}" for o, c in ops.most_common(24)))
