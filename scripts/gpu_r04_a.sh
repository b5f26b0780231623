#!/bin/bash
# round 4, first call: the ShardSync-on-HIP test, the sharding tests, the 2-rank rehearsal of bench.py, a default bench line
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
step 600 a_tests.log python -m pytest tests -m gpu -q -x -k "shard or parked or schedulers or full_size_wave"
TAILN=1 step 400 a_bench_default.json python bench.py
VSPG_BENCH_REHEARSE=1 TAILN=1 step 300 a_rehearse2.json python bench.py --gpus 2 --steps 6 --warmup 1 --no-pmc
TAILN=1 step 300 a_bench_cloud_guided.json python bench.py --workload cloud-guided --steps 8 --warmup 2 --train-waves 8 --no-cpu-baseline
exit 0
