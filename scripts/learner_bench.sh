#!/bin/bash
# Runs on the GPU box: ms per step of the deterministic learner workloads (cold K=356/355, warm K=356/355, K=1564/1563), HIP events.
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { timeout -k 10 200 python3 $R/bench.py --no-extra --no-cpu-baseline --steps 5 --warmup 2 "$@" | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print('%-24s K=%-5d %8.2f ms/step  %.3f M examples/s' % (r['config']['workload'], r['config']['classes'], r['roofline']['kernel_ms'], r['value']/1e6))" || exit 1; }
run --workload train_356sp_det && run --workload train_356sp_det --classes 355 && run --workload train_356sp_det_warm && run --workload train_356sp_det_warm --classes 355 && run --workload train_1564sp_det && run --workload train_1564sp_det --classes 1563
