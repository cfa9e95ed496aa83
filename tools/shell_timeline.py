#!/usr/bin/env python3
"""Per-shell timeline of the column sweep from a rocprofv3 --kernel-trace CSV (bench_kernel_trace.csv):
for the last complete pass, one line per k_sweep_shell / k_loss_* launch in stream order with its duration,
the gap to the previous launch and the algorithmic GB/s (88 B per cell.source).

    tools/shell_timeline.py gpurun_out/prof_x/stats/bench_kernel_trace.csv [--sources 8] [--out profiles/rNN_sweep_timeline.txt]
"""
import argparse
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--sources", type=int, default=8)
    ap.add_argument("--out")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        n = r["Kernel_Name"]
        kind = next((k for k in ("k_sweep_shell", "k_sweep_core", "k_loss_finish", "k_loss", "k_rates", "k_chemistry", "k_transpose_ij", "k_pack_state", "k_transpose_packed") if k in n), None)
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"])))
    rows.sort()
    # passes: runs of sweep/loss launches between two k_rates launches; take the last complete one
    # a pass starts with k_transpose_ij; the rates and chemistry launches are listed but not part of the sweep
    passes, cur = [], []
    for row in rows:
        if row[2] in ("k_transpose_ij", "k_pack_state", "k_transpose_packed"):
            if cur:
                passes.append(cur)
            cur = []
        elif row[2] != "k_chemistry":
            cur.append(row)
    if cur:
        passes.append(cur)
    p = passes[-1] if passes else []
    out = []
    t0 = p[0][0] if p else 0
    busy = 0
    shell = 0
    out.append(f"# last sweep pass of {a.csv}: {len(p)} launches")
    out.append(f"# {'kernel':14s} {'shell':>5s} {'blocks':>7s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'GB/s(88B)':>10s}")
    prev_end = None
    for s, e, kind, gx, gy in p:
        dur = (e - s) / 1e3
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = e
        busy += e - s
        rate = ""
        sh = ""
        if kind == "k_rates":
            busy -= e - s
        if kind.startswith("k_sweep"):
            cells = 1 if shell == 0 else 24 * shell * shell + 2
            rate = f"{88.0 * cells * gy / (e - s):10.1f}"
            sh = str(shell)
            shell += 1
        out.append(f"  {kind:14s} {sh:>5s} {gx // 256:7d} {(s - t0) / 1e3:9.1f} {dur:8.1f} {gap:7.1f} {rate:>10s}")
    if p:
        span = (p[-1][1] - t0) / 1e3
        out.append(f"# span {span:.1f} us, kernels busy {busy / 1e3:.1f} us, gaps {span - busy / 1e3:.1f} us")
    text = "\n".join(out) + "\n"
    sys.stdout.write(text)
    if a.out:
        open(a.out, "w").write(text)


if __name__ == "__main__":
    main()
