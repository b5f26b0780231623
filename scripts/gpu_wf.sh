#!/bin/bash
# wavefront pipeline bring-up: heterogeneous parity tests + cloud bench lines
mkdir -p gpurun_out
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-6} gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step 600 wf_tests.log python -m pytest tests -m gpu -q -x -k "${WF_K:-cloud or grid or nvdb or tr_buffer or emissive or full_size}"
TAILN=2 step 300 wf_bench_cloud.json python bench.py --workload cloud --steps 8 --warmup 2 --no-cpu-baseline
TAILN=2 step 300 wf_bench_cloud_nvdb.json python bench.py --workload cloud-nvdb --steps 8 --warmup 2 --no-cpu-baseline
exit 0
