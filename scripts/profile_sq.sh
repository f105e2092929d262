#!/bin/bash
# Runs on the GPU box (via gpurun): one SQ/GRBM PMC pass (issue/wait breakdown of the dominant kernel).
# usage: scripts/profile_sq.sh TAG [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc ${OPAL_PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra "$@" > $OUT/sq.log 2>&1 || { echo "SQ pass failed"; tail -5 $OUT/sq.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:3]:
    print(k, "launches", n[k])
    for c, x in sorted(v.items()): print("   %-22s %.4g per launch" % (c, x / max(n[k], 1)))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            print("   %-22s %.1f %% of wave cycles" % (c, 100 * v.get(c, 0) / wc))
PY
