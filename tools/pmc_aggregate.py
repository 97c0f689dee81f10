#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc output (counter_collection.csv) to bytes per launch per kernel.

  python tools/pmc_aggregate.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB by rocprofv3; on gfx950 FETCH_SIZE counts wide
streaming reads at half their bytes (MI355X_MICROARCH.md section HBM) and is doubled here; WRITE_SIZE is exact.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter):
    acc, n = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:96]
                acc[k] += float(row["Counter_Value"])
                n[k] += 1
    return acc, n


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        launches = max(nf.get(k, 0), nw.get(k, 0))
        fb = 2.0 * fetch.get(k, 0.0) * 1024 / max(nf.get(k, 1), 1)
        wb = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
        out[k] = {"launches": launches, "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": fb + wb}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 "
                    "--warmup 1`; FETCH_SIZE x2 (gfx950 correction), unit 1 KB = 1024 B; averages over all launches "
                    "of a kernel name in the run")
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]) if isinstance(kv[1], dict) else 0)[:12]:
        if isinstance(v, dict):
            print(f"{k[:60]:60s} n={v['launches']:5d} fetch={v['fetch_bytes_per_launch_corrected']/1e6:9.2f} MB write={v['write_bytes_per_launch']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
