#!/bin/bash
# Runs on the GPU box (via gpurun): three PMC passes (issue mix, LDS, scalar unit) over the cfg5 emit kernels.
# usage: scripts/emit_counters.sh TAG
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/scripts/profile_sq.sh ${TAG}_a --workload cfg5_emit > /dev/null || exit 1
OPAL_PMC="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" bash $R/scripts/profile_sq.sh ${TAG}_b --workload cfg5_emit > /dev/null || exit 1
OPAL_PMC="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU" bash $R/scripts/profile_sq.sh ${TAG}_c --workload cfg5_emit > /dev/null || exit 1
python3 - $R/gpurun_out $TAG <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
for f in glob.glob(root + "/sq_" + tag + "_*/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "emit_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        launches = max(n[k][c] // max(1, min(n[k].values())), 1) * min(n[k].values())
        print("   %-22s %.4g per launch" % (c, x / max(n[k][c], 1) * (1 if True else launches)))
PY
