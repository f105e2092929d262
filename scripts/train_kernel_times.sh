#!/bin/bash
# Runs on the GPU box: per-kernel times of the deterministic trainer at K = 1564 and K = 356 (first pass), kernel-trace only.
R=${GRAFT_REPO_ROOT:-/root/repo}
OPAL_PROF_KT_ONLY=1 OPAL_PROF_STEPS=3 bash $R/scripts/profile.sh tmp_train_1564 --workload train_1564sp_det "$@" 2>&1 | grep -E "dtrain_(apply|score|const)" | cut -c1-160
OPAL_PROF_KT_ONLY=1 OPAL_PROF_STEPS=3 bash $R/scripts/profile.sh tmp_train_356 --workload train_356sp_det "$@" 2>&1 | grep -E "dtrain_(apply|score|const)" | cut -c1-160
