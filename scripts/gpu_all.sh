#!/bin/bash
# the whole GPU suite + smoke + the bench lines (stops after a step that was killed)
mkdir -p gpurun_out
T=${1:-r05}
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} gpurun_out/$log | cut -c1-400; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
step 900 ${T}_tests_gpu.log python -m pytest tests -m gpu -q -x
step 200 ${T}_smoke.log python -c "import __graft_entry__ as g; g.smoke()"
TAILN=1 step 400 ${T}_bench_default.json python bench.py
TAILN=1 step 300 ${T}_bench_cloud.json python bench.py --workload cloud --steps 8 --warmup 2
TAILN=1 step 300 ${T}_bench_cloud_nvdb.json python bench.py --workload cloud-nvdb --steps 8 --warmup 2
TAILN=1 step 300 ${T}_bench_fog_guided.json python bench.py --workload fog-guided --steps 16 --warmup 2
TAILN=1 step 300 ${T}_bench_cloud_guided.json python bench.py --workload cloud-guided --steps 8 --warmup 2 --train-waves 8
TAILN=1 step 300 ${T}_bench_cloud_nvdb_guided.json python bench.py --workload cloud-nvdb-guided --steps 8 --warmup 2 --train-waves 8
TAILN=1 step 300 ${T}_bench_cloud_scene.json python bench.py --workload cloud-scene --steps 8 --warmup 2
TAILN=1 step 300 ${T}_bench_cloud_scene_nvdb.json python bench.py --workload cloud-scene-nvdb --steps 8 --warmup 2
TAILN=1 step 300 ${T}_bench_cloud_scene_guided.json python bench.py --workload cloud-scene-guided --steps 8 --warmup 2 --train-waves 8
exit 0
