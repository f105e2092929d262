#!/bin/bash
# Runs on the GPU box (via gpurun): the round's profiles of every bench.py workload DESIGN.md quotes.  usage: scripts/profile_round.sh r03
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1
set -e
bash $R/scripts/profile.sh ${T}_cfg2_356sp_1M --workload cfg2_356sp_1M > /dev/null
bash $R/scripts/profile.sh ${T}_cfg2_355sp --workload cfg2_356sp_1M --classes 355 > /dev/null
OPAL_PROF_STEPS=3 bash $R/scripts/profile.sh ${T}_cfg3_1564sp --workload cfg3_1564sp > /dev/null
bash $R/scripts/profile.sh ${T}_cfg3_1563sp --workload cfg3_1564sp --classes 1563 --nfrag 2000000 > /dev/null
bash $R/scripts/profile.sh ${T}_cfg4_numhash4 --workload cfg4_numhash4 > /dev/null
OPAL_PROF_STEPS=30 OPAL_PROF_WARMUP=10 bash $R/scripts/profile.sh ${T}_cfg5_emit --workload cfg5_emit > /dev/null
bash $R/scripts/emit_counters.sh ${T}cfg5 > $R/gpurun_out/${T}_cfg5_sq_lds_salu_counters.txt 2>&1
OPAL_PROF_STEPS=20 OPAL_PROF_WARMUP=5 bash $R/scripts/profile.sh ${T}_ingest --workload ingest_reads > /dev/null
OPAL_PROF_KT_ONLY=1 OPAL_PROF_STEPS=3 bash $R/scripts/profile.sh ${T}_train_356sp_det --workload train_356sp_det > /dev/null
OPAL_PROF_KT_ONLY=1 OPAL_PROF_STEPS=3 bash $R/scripts/profile.sh ${T}_train_355sp_det --workload train_356sp_det --classes 355 > /dev/null
OPAL_PROF_KT_ONLY=1 OPAL_PROF_STEPS=3 bash $R/scripts/profile.sh ${T}_train_1564sp_det --workload train_1564sp_det > /dev/null
OPAL_PROF_KT_ONLY=1 OPAL_PROF_STEPS=3 bash $R/scripts/profile.sh ${T}_train_1563sp_det --workload train_1564sp_det --classes 1563 > /dev/null
echo all profiles done
