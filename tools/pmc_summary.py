#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs of bench.py (one directory per counter set, collected in separate
passes as MI355X_MICROARCH.md prescribes) into a small JSON for profiles/.
usage: tools/pmc_summary.py out.json dir1 dir2 ..."""
import collections
import csv
import glob
import json
import sys


def kname(n):
    for k in ("k_rates", "k_chemistry", "k_sweep_shell_fast", "k_sweep_shell", "k_loss_finish_rounds", "k_loss_finish",
              "k_loss_probe_rounds", "k_loss_stored", "k_loss", "k_transpose_packed", "k_pack_state", "k_transpose_ij"):
        if k in n:
            return k
    return None


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in dirs:
        for f in glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = kname(r["Kernel_Name"])
                if k:
                    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    calls[k][r["Counter_Name"]] += 1
    res = {}
    for k in agg:
        res[k] = {c: {"total": v, "launches": calls[k][c], "per_launch": v / calls[k][c]} for c, v in agg[k].items()}
        if "FETCH_SIZE" in res[k] and "WRITE_SIZE" in res[k]:
            # units: KiB.  gfx950: FETCH_SIZE reports 1/2 of the bytes of coalesced streaming reads
            # (MI355X_MICROARCH.md "HBM"; confirmed here on __amd_rocclr_copyBuffer and on k_chemistry,
            # whose 8-byte-per-lane coalesced reads total 17 doubles per cell): doubled below.
            f, w = res[k]["FETCH_SIZE"]["per_launch"], res[k]["WRITE_SIZE"]["per_launch"]
            res[k]["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
        fl = [res[k].get("SQ_INSTS_VALU_%s_F64" % n) for n in ("FMA", "MUL", "ADD", "TRANS")]
        if all(fl):
            # wave-instructions; a wave64 instruction is 64 lane operations when every lane is active (an upper bound)
            fma, mul, add, trans = (x["per_launch"] for x in fl)
            res[k]["fp64_flop_per_launch_upper"] = 64.0 * (2.0 * fma + mul + add + trans)
    # which sources the counters were taken from (bench.py quotes them only for the same sources)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import bench
        res["source_sha16"] = bench.kernel_source_sha16()
    except Exception as ex:  # noqa: BLE001
        res["source_sha16"] = None
        sys.stderr.write(f"pmc_summary: no source hash ({ex})\n")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v.get("hbm_bytes_per_launch_corrected") for k, v in res.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
