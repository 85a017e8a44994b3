"""Summarises rocprofv3 --pmc passes (counter_collection.csv files) of `python bench.py ...` into the
JSON bench.py reads for roofline.traffic (profiles/pmc_<workload>_<precision>.json).

    python tools/summarize_pmc.py [--halves] OUT.json KERNEL_SUBSTR DIR_FETCH DIR_WRITE [DIR_SQ]

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies 64 B
per 128-B request (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact.  Averages are taken over
the dispatches of the named kernel only."""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def collect(d, kernel):
    acc = defaultdict(list)
    # this rocprofv3 writes a rocpd sqlite database by default (view counters_collection) ...
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        con = sqlite3.connect(f)
        for disp, name, cname, val in con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
            if kernel in name:
                per_dispatch[disp][cname] += float(val)
        for disp in per_dispatch.values():
            for k, v in disp.items():
                acc[k].append(v)
    # ... and counter_collection.csv with --output-format csv
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
        for disp in per_dispatch.values():
            for k, v in disp.items():
                acc[k].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def collect_halves(d, kernel):
    """FETCH_SIZE / WRITE_SIZE averages of the first and of the second half of the kernel's dispatches, in dispatch order
    (tools/time_esdf.py launches on uniformly random queries first, then on the same queries brick-sorted)"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        con = sqlite3.connect(f)
        for disp, name, cname, val in con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
            if kernel in name:
                per_dispatch[int(disp)][cname] += float(val)
        ids = sorted(per_dispatch)
        h = len(ids) // 2
        for label, part in (("first_half", ids[:h]), ("second_half", ids[h:])):
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                vals = [per_dispatch[i][c] for i in part if c in per_dispatch[i]]
                if vals:
                    out.setdefault(label, {})[c + "_KiB_raw"] = sum(vals) / len(vals)
    return out


def kernel_stats_csv(trace_dir, out_csv):
    """the --stats table (name, calls, total ns, average ns, %) of a --kernel-trace run, as CSV"""
    for f in glob.glob(os.path.join(trace_dir, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        with open(out_csv, "w", newline="") as fh:
            w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for name, n, total, avg, mn, mx in rows:
                w.writerow([name, n, total, round(avg, 3), round(100.0 * total / tot, 4), mn, mx])
        return rows
    return []


def main():
    if sys.argv[1] == "--kernel-stats":
        for r in kernel_stats_csv(sys.argv[2], sys.argv[3])[:5]:
            print(r)
        return
    halves = "--halves" in sys.argv
    if halves:
        sys.argv.remove("--halves")
    out, kernel, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    merged = {}
    for d in dirs:
        merged.update(collect(d, kernel))
    fetch_kib = merged.get("FETCH_SIZE", (0.0, 0))[0]
    write_kib = merged.get("WRITE_SIZE", (0.0, 0))[0]
    try:    # the library's own identity: content hashes of its sources (include/vigo.h vigo_build_id; loading it starts no GPU work)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from trajectory_planner_amd import _lib
        build_id = _lib.load().vigo_build_id().decode()
    except Exception:
        build_id = None
    res = {
        "build_id": build_id,
        "build": os.environ.get("VIGO_BUILD", "unrecorded"),   # commit the profiled library was built from (set by the caller)
        "source": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python bench.py (see profiles/README.md); "
                  f"averages over the dispatches of the kernel matching '{kernel}'",
        "dispatches": {k: n for k, (_, n) in merged.items()},
        "FETCH_SIZE_KiB_raw": fetch_kib,
        "WRITE_SIZE_KiB_raw": write_kib,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section; an upper "
                      "bound for this kernel's narrow reads); WRITE_SIZE taken as is",
        "hbm_bytes_per_launch": int(round((2.0 * fetch_kib + write_kib) * 1024.0)),
        "counters_per_launch": {k: v for k, (v, _) in merged.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")},
    }
    if halves:
        res["split"] = {}
        for d in dirs:
            for label, vals in collect_halves(d, kernel).items():
                res["split"].setdefault(label, {}).update(vals)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
