#!/usr/bin/env python3
"""Dumps the per-kernel statistics of a rocprofv3 (ROCm 7.x, rocpd sqlite output) run as CSV — the
same table `rocprofv3 --stats` prints (`top_kernels` view): name, calls, total us, average us, %.

    python tools/rocpd_summary.py gpurun_out/.../bench_results.db > profiles/rNN/kernel_stats.csv
"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])   # the top_kernels view reports microseconds
for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    w.writerow([name[:160], calls, int(total), int(avg), round(pct, 3)])
try:
    kcols = [c[1] for c in con.execute("pragma table_info('kernel_symbols')")]
    namecol = "display_name" if "display_name" in kcols else ("kernel_name" if "kernel_name" in kcols else kcols[1])
    rows = list(con.execute(f"select k.{namecol}, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, d.group_segment_size, "
                            "d.private_segment_size, d.end - d.start from rocpd_kernel_dispatch d "
                            "join kernel_symbols k on k.id = d.kernel_id order by d.start"))
    w.writerow([])
    w.writerow(["dispatch", "grid_x", "grid_y", "wg_x", "lds_bytes", "scratch_bytes_per_lane", "duration_ns"])
    for r in rows[-12:]:
        w.writerow([str(r[0])[:100]] + list(r[1:]))
except Exception as e:  # noqa: BLE001
    w.writerow(["(dispatch list unavailable: %s)" % e])
