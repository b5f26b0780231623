#!/bin/bash
# round 5: first run of the barrier-free scheduler (k_render_wave_wg3): scheduler tests under a short timeout, then headline A/B
mkdir -p gpurun_out/r05
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -q -x -k "schedulers_are_bit_identical or parked_samples" > gpurun_out/r05/a_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r05/a_tests.log
[ $rc -eq 0 ] || exit $rc
for sched in 3 2 3 2; do
  VSPG_WG_SCHED=$sched timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-generic --no-reference-defaults > gpurun_out/r05/a_bench_s$sched.json 2>gpurun_out/r05/a_bench_s$sched.err || { echo "bench sched $sched failed"; tail -3 gpurun_out/r05/a_bench_s$sched.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05/a_bench_s$sched.json").read().strip().split("\n")[-1])
print("sched $sched: value %.1f ms_per_step %.4f kernel_ms %.4f kernel %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel"]))
PY
done
