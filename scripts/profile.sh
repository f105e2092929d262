#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + the two PMC passes for HBM traffic, each in its own run.
# usage: [OPAL_PROF_STEPS=5 OPAL_PROF_WARMUP=1] scripts/profile.sh TAG [bench args...]
# OPAL_PROF_KT_ONLY=1: the kernel-trace run only (workloads whose traffic is not quoted)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps ${OPAL_PROF_STEPS:-5} --warmup ${OPAL_PROF_WARMUP:-1} --no-cpu-baseline --no-extra "$@" > $OUT/kt.log 2>&1 || { echo "kernel-trace run failed"; tail -5 $OUT/kt.log; exit 1; }
if [ -z "$OPAL_PROF_KT_ONLY" ]; then
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra "$@" > $OUT/fetch.log 2>&1 || { echo "FETCH_SIZE run failed"; tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra "$@" > $OUT/write.log 2>&1 || { echo "WRITE_SIZE run failed"; tail -5 $OUT/write.log; exit 1; }
fi
python3 $R/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
