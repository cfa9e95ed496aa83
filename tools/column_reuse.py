#!/usr/bin/env python3
"""How much of a pass repeats the previous outer iteration bit for bit?  (reuse of columns and
rate contributions across outer iterations is only open to a bit-exact path for cell.source pairs whose columns are
IDENTICAL to the previous iteration's.)

Per outer iteration and per sampled source: the source's outgoing columns N_out(HI, HeI, HeII) of every cell
(c2r_download_columns after c2r_do_source) compared with the same source's columns of the previous iteration; a
cell.source pair counts as "unchanged" when all three are bit-identical.  (The incoming columns are interpolated from
the N_out of four upstream cells, so this is an UPPER bound for "all six columns identical".)  The physics runs as
always: after the census the rate grids are cleared and the whole pass is done by c2r_pass_sources.

    tools/column_reuse.py --workload bench   [--mesh 256] [--max-iter 60]     bench.py --neutral-start: 8 bright sources
    tools/column_reuse.py --workload config3 [--mesh 256] [--sample 16]       1024 faint sources, log-normal density
Prints one JSON line (copy to profiles/rNN_column_reuse.json)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["bench", "config3"], default="bench")
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--max-iter", type=int, default=60)
    ap.add_argument("--sample", type=int, default=8, help="sources whose columns are compared (evenly spaced in the list)")
    ap.add_argument("--calls", type=int, default=1, help="evolve3D calls (time steps) in a row")
    a = ap.parse_args()
    pkg = ge.load_package()
    n = a.mesh
    if a.workload == "bench":
        mat, grid, src, cosmo = bench.config3_inputs(pkg, n, 8, neutral=True)
        batch = 8
    else:
        mat, grid, src, cosmo = bench.config4_inputs(pkg, n, 1024)
        batch = 256
    nsrc = src.NumSrc
    sample = sorted(set(int(x) for x in np.linspace(1, nsrc, min(a.sample, nsrc)).round()))
    e = pkg.HipEngine((n, n, n), 0)
    e.set_tables(pkg.RadiationTables.load())
    e.set_batch(batch)
    dt = 1.0e7 * pkg.hostphys.YEAR
    nc = n ** 3
    conv_criterion = min(int(2.5e-4 * nc), nsrc)
    calls = []
    self_check = False
    for call in range(a.calls):
        e.set_step(mat, grid, cosmo)
        e.set_sources(src)
        e.upload_state(mat)
        e.begin_step()
        prev = {}
        rows = []
        niter, conv = 0, nc
        while not (conv < conv_criterion and niter > 1) and niter < a.max_iter:
            niter += 1
            # census: columns of the sampled sources as this iteration's pass makes them
            e.set_rates_to_zero()
            same = total = 0
            per_src, rel_changes = [], []
            for ns in sample:
                e.do_source(ns)
                c = e.download_columns()
                now = np.stack([c["coldensh_out"], c["coldenshe_out"][:nc], c["coldenshe_out"][nc:]])
                traced = now[0] != 0.0           # cells outside the source's last sub-box hold 0
                if ns == sample[0] and niter == 2:
                    # the instrument itself: the same source traced twice from the same state is identical everywhere
                    e.do_source(ns)
                    c2 = e.download_columns()
                    again = np.stack([c2["coldensh_out"], c2["coldenshe_out"][:nc], c2["coldenshe_out"][nc:]])
                    assert np.array_equal(again, now), "a source traced twice from one state differs"
                    self_check = True
                if ns in prev:
                    eq = (now.view(np.int64) == prev[ns].view(np.int64)).all(axis=0) & traced
                    same += int(eq.sum())
                    total += int(traced.sum())
                    per_src.append(float(eq.sum()) / max(1, int(traced.sum())))
                    both = traced & (prev[ns][0] != 0.0)
                    rel = np.abs(now[0][both] - prev[ns][0][both]) / prev[ns][0][both]
                    rel_changes.append((float(np.median(rel)), float(rel.min()), float(rel.max())))
                prev[ns] = now
            # the iteration itself
            e.set_rates_to_zero()
            e.pass_sources(1, 1)
            conv = e.global_pass(dt)
            rows.append({"iter": niter, "nonconv": int(conv), "pairs_compared": total, "unchanged_fraction": (same / total) if total else None,
                         "per_source_min_max": [min(per_src), max(per_src)] if per_src else None,
                         # relative change of N_out(HI) against the previous iteration over the cells of a source: median, least, largest
                         "rel_change_NoutHI_median_min_max": [float(np.median([r[0] for r in rel_changes])), min(r[1] for r in rel_changes),
                                                             max(r[2] for r in rel_changes)] if rel_changes else None})
            sys.stderr.write(f"call {call + 1} iteration {niter}: nonconv {conv}, unchanged {rows[-1]['unchanged_fraction']}\n")
        e.end_step()
        e.download_state(mat)
        w = [(r["pairs_compared"], r["unchanged_fraction"]) for r in rows if r["unchanged_fraction"] is not None]
        calls.append({"call": call + 1, "iterations": niter,
                      "pair_weighted_unchanged_fraction": sum(p * f for p, f in w) / max(1, sum(p for p, _ in w)),
                      "per_iteration": rows})
    print(json.dumps({"workload": a.workload, "mesh": n, "sources": nsrc, "sampled_sources": sample,
                      "instrument_self_check_passed": self_check, "what": "fraction of traced cell.source pairs whose three outgoing columns are bit-identical to the previous "
                              "outer iteration's (upper bound for all six columns identical)", "calls": calls}))
    e.close()


if __name__ == "__main__":
    main()
