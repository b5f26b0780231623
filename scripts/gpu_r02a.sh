#!/bin/bash
# round 2, first GPU contact: issue-rate microbenchmark, parity tests, smoke, the bench lines.  Stops at the first
# step that times out (no further GPU step after a kill).
mkdir -p gpurun_out
step() {  # step <seconds> <log> <cmd...>
  local t=$1 log=$2; shift 2
  timeout -k 10 $t "$@" > gpurun_out/$log 2>&1
  local rc=$?
  echo "$log rc=$rc"; tail -4 gpurun_out/$log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
step 120 r02_microbench_issue.txt scripts/microbench/imul
step 600 r02a_tests_gpu.log python -m pytest tests -m gpu -q -s -x
step 200 r02a_smoke.log python -c "import __graft_entry__ as g; g.smoke()"
step 400 r02a_bench_default.json python bench.py
step 300 r02a_bench_cloud.json python bench.py --workload cloud --steps 8 --warmup 2
step 300 r02a_bench_cloud_nvdb.json python bench.py --workload cloud-nvdb --steps 8 --warmup 2
step 300 r02a_bench_fog_guided.json python bench.py --workload fog-guided --steps 16 --warmup 2
exit 0
