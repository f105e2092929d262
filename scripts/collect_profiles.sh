#!/bin/bash
# Copies what scripts/profile_round.sh left under gpurun_out/ into profiles/<tag>_<workload>/ (the files DESIGN.md cites) and
# regenerates profiles/traffic.json.  usage: scripts/collect_profiles.sh r03
T=$1
cd "$(dirname "$0")/.."
for d in gpurun_out/prof_${T}_*; do
  [ -d $d ] || continue
  t=${d#gpurun_out/prof_}
  o=profiles/$t
  mkdir -p $o
  cp $(ls -t $d/kt/*/*_kernel_stats.csv | head -1) $o/kernel_stats.csv      # (gpurun_out/ keeps earlier calls' files: the newest)
  cp $d/summary.txt $o/summary.txt
  if [ -d $d/fetch ]; then
    cp $(ls -t $d/fetch/*/*_counter_collection.csv | head -1) $o/fetch_counter_collection.csv
    cp $(ls -t $d/write/*/*_counter_collection.csv | head -1) $o/write_counter_collection.csv
  fi
done
[ -f gpurun_out/${T}_cfg5_sq_lds_salu_counters.txt ] && cp gpurun_out/${T}_cfg5_sq_lds_salu_counters.txt profiles/${T}_cfg5_emit/sq_lds_salu_counters.txt
python3 scripts/make_traffic_json.py > /dev/null
