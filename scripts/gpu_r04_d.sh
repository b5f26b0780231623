#!/bin/bash
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-900; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=12 step 400 d_scenefile.log python -m pytest tests/test_host_adapter.py -m gpu -q -x -s -k "boundaries or cloud_sky or equals_the_api"
TAILN=1 step 300 d_bench_cloud_scene.json python bench.py --workload cloud-scene --steps 8 --warmup 2 --no-cpu-baseline
TAILN=1 step 300 d_bench_cloud.json python bench.py --workload cloud --steps 8 --warmup 2 --no-cpu-baseline
TAILN=1 step 300 d_bench_cloud_scene_guided.json python bench.py --workload cloud-scene-guided --steps 8 --warmup 2 --train-waves 8 --no-cpu-baseline
TAILN=1 step 300 d_bench_cloud_scene_nvdb.json python bench.py --workload cloud-scene-nvdb --steps 8 --warmup 2 --no-cpu-baseline
exit 0
