#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into the text summary that is
committed under profiles/.  usage: rocprof_summary.py <stats_dir> [<pmc_dir> ...]"""
import csv, glob, sys, collections

def main():
    stats_dir, pmc_dirs = sys.argv[1], sys.argv[2:]
    for f in glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True):
        print("== kernel stats (rocprofv3 --kernel-trace --stats):", f.split("/")[-1])
        print(f"{'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  {'%':>6}  name")
        for r in csv.DictReader(open(f)):
            print(f"{r['Calls']:>6} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} "
                  f"{float(r['MaxNs'])/1e3:10.2f}  {float(r['Percentage']):6.2f}  {r['Name'][:110]}")
    for d in pmc_dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[(r["Kernel_Name"][:72], r["Counter_Name"])].append(float(r["Counter_Value"]))
            print("== PMC (rocprofv3 --pmc):", d)
            for (k, c), v in sorted(agg.items()):
                if "va::" not in k and "_ZN2va" not in k:
                    continue
                avg = sum(v) / len(v)
                extra = ""
                if c == "FETCH_SIZE":
                    extra = f"  -> x2 (gfx950 correction) = {avg * 2 * 1024 / 1e6:.1f} MB read / launch"
                if c == "WRITE_SIZE":
                    extra = f"  -> {avg * 1024 / 1e6:.2f} MB written / launch"
                unit = "(KB)" if c in ("FETCH_SIZE", "WRITE_SIZE") else ""
                print(f"  {c:24s} n={len(v):3d} avg={avg:14.1f} {unit}{extra}   {k}")

if __name__ == "__main__":
    main()
