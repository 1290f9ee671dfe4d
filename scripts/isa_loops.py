"""Development aid: per-loop instruction mix of one kernel in an `hipcc -S --cuda-device-only` listing.
usage: isa_loops.py listing.s mangled_kernel_name_substring"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(";")[0].strip().endswith(":"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].strip().startswith(".end_amdhsa_kernel") or (lines[i].startswith("_Z") and lines[i].endswith(":")))
body = lines[start:end]
labels, ins = {}, []
for l in body:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not t or t.startswith(".") or t.startswith(";") or t.endswith(":"):
        continue
    ins.append(t.split(";")[0].strip())


def kind(i):
    op = i.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"


print(f"{key}: {len(ins)} instructions", Counter(kind(i) for i in ins))
loops = []
for idx, i in enumerate(ins):
    m = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", i)
    if m and m.group(1) in labels and labels[m.group(1)] <= idx:
        loops.append((labels[m.group(1)], idx, m.group(1)))
loops.sort()
for lo, hi, lab in loops:
    c = Counter(kind(i) for i in ins[lo:hi + 1])
    ops = Counter(i.split()[0] for i in ins[lo:hi + 1] if kind(i) == "valu")
    print(f"  loop {lab} [{lo}..{hi}] n={hi - lo + 1} {dict(c)}  top valu: {ops.most_common(8)}")
