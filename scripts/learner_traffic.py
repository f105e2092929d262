#!/usr/bin/env python3
"""profiles/<tag>_train_*_det/learner_traffic.txt from the summary.txt scripts/profile.sh left (full run, with the PMC passes).
usage: scripts/learner_traffic.py profiles/r03_train_356sp_det"""
import json
import os
import sys

d = sys.argv[1]
s = open(os.path.join(d, "summary.txt")).read()
j = json.loads(s[s.index("== summary ==") + len("== summary =="):])
out = ["# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (2 steps each), per dispatch, against the mean duration of the",
       "# --kernel-trace run of the same command (scripts/profile.sh).  WRITE_SIZE is exact for 16-B-per-lane stores (guide, HBM section).",
       "# An apply launch reads every table row it writes (G and w, read-modify-write), so its table traffic = 2 x WRITE_SIZE; the rest of",
       "# its corrected reads (FETCH_SIZE x 2, the gfx950 wide-read correction) are the group's step arrays upd/gsq (23 MB at K=356, 102 MB",
       "# at K=1564: re-read per row out of the Infinity Cache, which the counter includes).  The score kernel's corrected reads are its two",
       "# read-only sweeps over w and G rows."]
for k, v in j["learner_kernels"].items():
    if v["avg_ms"] > 0.1:
        out.append("%-38s launches %4d  avg %7.3f ms  FETCH_SIZE x2 %8.1f MB  WRITE_SIZE %8.1f MB  | table RMW (2 x writes) %6.0f GB/s | reads x2 + writes %6.0f GB/s"
                   % (k, v["calls"], v["avg_ms"], 2 * v["fetch_bytes_raw"] / 1e6, v["write_bytes"] / 1e6, v["GBps_floor_2x_writes"], v["GBps_reads_x2_plus_writes"]))
open(os.path.join(d, "learner_traffic.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
