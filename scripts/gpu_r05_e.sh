#!/bin/bash
# round 5: Field::Update as one graph launch: training tests, then the fog-guided training-wave timing with / without the graph
mkdir -p gpurun_out/r05
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "train or guided or schedulers or parked" > gpurun_out/r05/e_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r05/e_tests.log
[ $rc -eq 0 ] || exit $rc
for g in 1 0 1 0; do
  VSPG_TRAIN_GRAPH=$g timeout -k 10 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pmc --no-generic --no-fast-arith > gpurun_out/r05/e_bench_g$g.json 2>gpurun_out/r05/e_bench_g$g.err || { echo "bench failed"; tail -3 gpurun_out/r05/e_bench_g$g.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05/e_bench_g$g.json").read().strip().split("\n")[-1])
rd=d["reference_defaults"]
print("graph $g: headline %.1f; trained wave %.4f ms (%.1f Mpaths/s); training wave %.4f ms; effective_256spp %.1f" % (d["value"], rd["ms_per_trained_wave"], rd["value"], rd["training"]["ms_per_wave"], rd["effective_256spp"]["value"]))
PY
done
