#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs (one or more passes) as JSON.

    python tools/pmc_summary.py out.json pass1/**/counter_collection.csv pass2/... [--last-calls N]

Counters are summed over the LAST N dispatches of each kernel (default 1: the steady-state launch) so
that a warm-up launch does not double the figures.  Kernel names are shortened to the function name."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:60]


def main() -> None:
    args = sys.argv[1:]
    last = 1
    if "--last-calls" in args:
        i = args.index("--last-calls")
        last = int(args[i + 1])
        del args[i:i + 2]
    out, files = args[0], args[1:]
    result = defaultdict(dict)
    for f in files:
        per = defaultdict(lambda: defaultdict(dict))       # kernel -> counter -> dispatch id -> value
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                per[k][row["Counter_Name"]][int(row["Dispatch_Id"])] = \
                    per[k][row["Counter_Name"]].get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
        for k, counters in per.items():
            for c, by_dispatch in counters.items():
                ids = sorted(by_dispatch)[-last:]
                result[k][c] = sum(by_dispatch[i] for i in ids) / len(ids)
                result[k].setdefault("_dispatches_seen", len(by_dispatch))
    json.dump(result, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {c: v for c, v in d.items() if c in ("FETCH_SIZE", "WRITE_SIZE")} for k, d in result.items()}))


main()
