#!/usr/bin/env python3
"""Condenses a scripts/profile.sh output directory into the few numbers DESIGN.md / bench.py quote.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide (16 B/lane) coalesced read, so the read side is doubled for the
float4 row gathers; WRITE_SIZE is exact.  One --pmc pass per counter.
"""
import csv
import glob
import json
import os
import sys


def rows(pattern):
    for p in glob.glob(pattern, recursive=True):
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def main(out):
    res = {}
    stats = {}
    for r in rows(os.path.join(out, "kt", "**", "*kernel_stats.csv")):
        stats[r["Name"]] = r
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        print("%-90s calls %5s  avg %12.1f us  total %10.3f ms  %5s%%" % (
            name[:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    hot = [n for n in stats if "predict_kernel" in n or "emit_canon2_kernel" in n or "emit_idx_kernel<true>" in n or "train_kernel" in n]
    if hot:
        res["kernel"] = hot[0]
        res["avg_kernel_ms"] = float(stats[hot[0]]["AverageNs"]) / 1e6
        res["calls"] = int(stats[hot[0]]["Calls"])
    for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        vals = []
        for r in rows(os.path.join(out, sub, "**", "*counter_collection.csv")):
            if r.get("Counter_Name") == ctr and any(k in r["Kernel_Name"] for k in ("predict_kernel", "emit_canon2_kernel", "emit_idx_kernel<true>", "train_kernel")):
                vals.append(float(r["Counter_Value"]))
        if vals:
            res[ctr + "_KiB_per_launch"] = sum(vals) / len(vals)
            res[ctr + "_launches"] = len(vals)
    if not hot and any("ing_fused_kernel" in n for n in stats):
        # FASTA ingest: several kernels per call; counters are summed over them and divided by the number of calls
        res["kernel"] = "ing_fused_kernel + ing_edges_kernel x2 + ing_records_kernel"
        calls = int(next(r for n, r in stats.items() if "ing_fused_kernel" in n)["Calls"])
        res["calls"] = calls
        res["avg_kernel_ms"] = sum(float(r["TotalDurationNs"]) for n, r in stats.items() if "::ing_" in n) / calls / 1e6
        for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            tot, launches = 0.0, 0
            for r in rows(os.path.join(out, sub, "**", "*counter_collection.csv")):
                if r.get("Counter_Name") == ctr and "::ing_" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"])
                    launches += "ing_fused_kernel" in r["Kernel_Name"]
            if launches:
                res[ctr + "_KiB_per_launch"] = tot / launches
                res[ctr + "_launches"] = launches
    if any("dtrain_" in n for n in stats):
        # deterministic learner: several kernels per group; per kernel, mean counter values per dispatch against the mean
        # duration of the kernel-trace run.  Reads are quoted both ways (raw, and x2 as for the wide row gathers); writes are
        # exact, and since every row the apply kernels write they have also read, 2 x WRITE_SIZE is a floor for their traffic.
        per = {}
        for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            for r in rows(os.path.join(out, sub, "**", "*counter_collection.csv")):
                if r.get("Counter_Name") == ctr and "dtrain_" in r["Kernel_Name"]:
                    short = r["Kernel_Name"].split("dtrain_")[1].split("(")[0]
                    per.setdefault("dtrain_" + short, {}).setdefault(ctr, []).append(float(r["Counter_Value"]))
        tab = {}
        for n, r in stats.items():
            if "dtrain_" not in n:
                continue
            short = "dtrain_" + n.split("dtrain_")[1].split("(")[0]
            c = per.get(short, {})
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            ms = float(r["AverageNs"]) / 1e6
            f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024
            w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
            tab[short] = {"calls": int(r["Calls"]), "avg_ms": ms, "fetch_bytes_raw": f, "write_bytes": w,
                          "GBps_reads_x2_plus_writes": (2 * f + w) / ms / 1e6, "GBps_floor_2x_writes": 2 * w / ms / 1e6,
                          "GBps_raw_reads_plus_writes": (f + w) / ms / 1e6}
        if tab:
            res["learner_kernels"] = tab
    if "FETCH_SIZE_KiB_per_launch" in res:
        rd = res["FETCH_SIZE_KiB_per_launch"] * 1024 * 2   # gfx950: wide coalesced reads are tallied at half
        wr = res.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
        res["hbm_bytes_per_launch"] = rd + wr
        res["hbm_read_bytes_per_launch_corrected_x2"] = rd
        res["hbm_write_bytes_per_launch"] = wr
    print("== summary ==")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
