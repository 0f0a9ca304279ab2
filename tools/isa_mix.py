#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a compiler assembly listing (hipcc -S --cuda-device-only).
usage: isa_mix.py listing.s 'mangled-name substring'  -> every backward branch with its body's instruction groups"""
import collections, re, sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = end = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m and pat in m.group(1) and start is None:
        start = i
    if start is not None and l.startswith(".Lfunc_end"):
        end = i
        break
body = lines[start:end]
labels = {}
ins = []
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith((".", ";")):
        ins.append(t)
def grp(mn):
    if mn.startswith("v_mad_u64"): return "mad64"
    if mn.startswith("v_cndmask"): return "cndmask"
    if mn.startswith("v_mov") or mn.startswith("v_accvgpr"): return "mov"
    if "dpp" in mn: return "dpp"
    if mn.startswith("v_"): return "valu_other"
    if mn.startswith("ds_"): return "lds"
    if mn.startswith("s_waitcnt"): return "waitcnt"
    if mn.startswith("s_"): return "salu"
    if mn.startswith(("scratch_", "buffer_", "global_", "flat_")): return "vmem:" + mn.split("_")[0]
    return mn
print("kernel instructions:", len(ins))
loops = []
for i, t in enumerate(ins):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i))
for a, b in sorted(loops):
    c = collections.Counter(grp(t.split()[0] + (" dpp" if "dpp" in t else "")) for t in ins[a:b + 1])
    dpp = sum(1 for t in ins[a:b + 1] if "row_shr" in t or "quad_perm" in t or "row_bcast" in t or "_dpp" in t)
    print(f"loop [{a}..{b}] n={b - a + 1} dpp={dpp}", dict(c.most_common()))
