#!/bin/bash
# round 5: the tolerance-mode instantiations: their tests, then bench lines with the fast_arith record
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_fast_arith.py -q -x -s > gpurun_out/r05/d_tests.log 2>&1; rc=$?
grep -E "relMSE|segment counts|passed|failed|Error|error" gpurun_out/r05/d_tests.log | cut -c1-330
[ $rc -eq 0 ] || { tail -30 gpurun_out/r05/d_tests.log; exit $rc; }
for wl in fog cloud cloud-scene; do
  steps=40; case $wl in cloud*) steps=8;; esac
  timeout -k 10 400 python bench.py --workload $wl --steps $steps --warmup 4 --no-cpu-baseline --no-pmc --no-generic --no-reference-defaults > gpurun_out/r05/d_bench_$wl.json 2>gpurun_out/r05/d_bench_$wl.err || { echo "bench $wl failed"; tail -5 gpurun_out/r05/d_bench_$wl.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05/d_bench_$wl.json").read().strip().split("\n")[-1])
print("$wl: value %.1f ms_per_step %.4f kernel_ms %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
for k,v in d.get("fast_arith",{}).items():
    if isinstance(v, dict): print("   %-13s kernel_ms %.4f ms_per_step %.4f relmse_vs_exact %s flipped %s same_seg %s" % (k, v["kernel_ms"], v["ms_per_step"], v.get("relmse_vs_exact"), v.get("flipped_path_frac"), v.get("same_segment_count_frac")))
PY
done
