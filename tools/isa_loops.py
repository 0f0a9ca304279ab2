#!/usr/bin/env python3
"""Loop structure and instruction mix of one kernel in an llvm-objdump -d listing.

usage: isa_loops.py listing.s 'kernel name substring'
Prints every backward branch (a loop) with its byte size and the instruction mix of its body
(innermost-only counts are not separated: nested loops are listed individually)."""
import collections
import re
import subprocess
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    insts = []          # (addr, mnemonic, text)
    for ln in lines:
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", ln)
        if m:
            if start is not None:
                break
            name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            if pat in name:
                start = int(m.group(1), 16)
                print("kernel:", name[:120])
            continue
        if start is None:
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):", ln)
        if m:
            insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
    print("instructions:", len(insts), "bytes:", insts[-1][0] - insts[0][0])
    loops = []
    for i, (a, mn, txt) in enumerate(insts):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"(-?\d+)\s*$", txt.strip())
            if not m:
                continue
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_index:
                loops.append((addr_index[tgt], i))
    for (s, e) in sorted(loops):
        body = insts[s:e + 1]
        mix = collections.Counter(mn for _, mn, _ in body)
        nbytes = body[-1][0] - body[0][0]
        print(f"\nloop insts [{s}..{e}] n={len(body)} bytes={nbytes}")
        groups = collections.Counter()
        for mn, c in mix.items():
            if mn.startswith("v_mad_u64"):
                g = "mad64"
            elif mn.startswith("v_accvgpr"):
                g = "accvgpr"
            elif mn.startswith("v_"):
                g = "valu_other"
            elif mn.startswith("ds_"):
                g = "lds"
            elif mn.startswith("s_waitcnt"):
                g = "s_waitcnt"
            elif mn.startswith("s_nop"):
                g = "s_nop"
            elif mn.startswith("s_load") or mn.startswith("s_buffer"):
                g = "smem"
            elif mn.startswith("s_"):
                g = "salu"
            elif mn.startswith("scratch_") or mn.startswith("buffer_") or mn.startswith("global_") or mn.startswith("flat_"):
                g = "vmem:" + mn.split("_")[0] + "_" + mn.split("_")[1]
            else:
                g = mn
            groups[g] += c
        print("  groups:", dict(groups.most_common()))
        print("  top:", mix.most_common(14))


main()
