#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel totals and, for the column sweep,
busy time versus gaps on its stream.  usage: tools/trace_timeline.py results.db"""
import collections
import sqlite3
import statistics
import sys

c = sqlite3.connect(sys.argv[1])
ks = {r[0]: r[1] for r in c.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
rows = list(c.execute("select kernel_id, start, end, stream_id, grid_size_x from rocpd_kernel_dispatch order by start"))
agg = collections.defaultdict(list)
for kid, s, e, sid, gx in rows:
    name = ks[kid]
    for key in ("k_sweep_shell", "k_rates", "k_chemistry", "k_loss_finish", "k_transpose_ij", "k_col_to_grid", "k_state", "k_total", "k_stat"):
        if key in name:
            name = key
            break
    agg[name[:40]].append((e - s) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:42s} n={len(v):6d} total={sum(v)/1e3:9.3f} ms mean={statistics.mean(v):9.1f} us max={max(v):9.1f} us")
# sweep stream: passes are separated by k_rates launches
sw = [(s, e, gx) for kid, s, e, sid, gx in rows if "k_sweep_shell" in ks[kid] or "k_loss_finish" in ks[kid]]
if sw:
    passes, cur = [], [sw[0]]
    for a, b in zip(sw, sw[1:]):
        if b[0] - a[1] > 5e6:  # > 5 ms apart: next pass
            passes.append(cur)
            cur = []
        cur.append(b)
    passes.append(cur)
    for p in passes[-3:]:
        busy = sum(e - s for s, e, _ in p) / 1e6
        span = (p[-1][1] - p[0][0]) / 1e6
        gaps = sorted(((b[0] - a[1]) / 1e3 for a, b in zip(p, p[1:])), reverse=True)
        print(f"sweep pass: {len(p)} launches, span {span:.3f} ms, kernels busy {busy:.3f} ms, gaps {span-busy:.3f} ms; "
              f"largest gaps (us): {[round(g,1) for g in gaps[:15]]}; median gap {statistics.median(gaps):.1f} us")
        big = sorted(((e - s) / 1e3, gx) for s, e, gx in p)[-16:]
        print("   longest launches (us, grid):", [(round(d, 1), g) for d, g in big])
