#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small JSON / CSV summaries kept under profiles/rNN/.

    python3 profiles/summarize.py stats  <rocprof_dir> <out.csv>              # --kernel-trace --stats
    python3 profiles/summarize.py pmc    <out.json> <note> <dir> [<dir> ...]  # --pmc passes

`pmc` averages every counter per kernel over its dispatches.  FETCH_SIZE / WRITE_SIZE are reported
by rocprofv3 in KB; on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section), so `hbm_read_bytes` = 2 * 1024 * FETCH_SIZE; WRITE_SIZE is
exact for 16-byte streaming stores.  SQ_* cycle counters are in quad-cycles (same guide)."""
import csv
import glob
import json
import os
import subprocess
import sys


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def stats(d, out):
    rows = []
    for f in find(d, "*kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    keys = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    with open(out, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=keys, quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in keys})


def pmc(out, note, dirs):
    acc = {}
    for d in dirs:
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], [])
                k.append(float(r["Counter_Value"]))
    kernels = {}
    for name, counters in sorted(acc.items()):
        e = {}
        for c, vals in sorted(counters.items()):
            e[c + "_avg"] = sum(vals) / len(vals)
            e["dispatches_" + c] = len(vals)
        if "FETCH_SIZE_avg" in e:
            e["hbm_read_bytes"] = 2.0 * 1024.0 * e["FETCH_SIZE_avg"]
        if "WRITE_SIZE_avg" in e:
            e["hbm_write_bytes"] = 1024.0 * e["WRITE_SIZE_avg"]
        if "SQ_BUSY_CYCLES_avg" in e and "SQ_ACTIVE_INST_VALU_avg" in e and e.get("SQ_WAVE_CYCLES_avg"):
            e["valu_active_share_of_wave_cycles"] = e["SQ_ACTIVE_INST_VALU_avg"] / e["SQ_WAVE_CYCLES_avg"]
        kernels[name] = e
    # the build that was profiled: handed over by the recipe (the GPU box has no .git), else HEAD here
    commit = os.environ.get("GRAVHMC_PROFILED_COMMIT")
    if not commit:
        try:
            commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True,
                                             cwd=os.path.dirname(os.path.abspath(__file__))).strip()
        except Exception:
            commit = "n/a"
    json.dump({"note": note, "commit": commit, "kernels": kernels}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4:])
