#!/bin/bash
# round 5: instruction counts of the two schedulers (bench.py's own PMC child passes)
mkdir -p gpurun_out/r05
for sched in 3 2; do
  VSPG_WG_SCHED=$sched timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-generic --no-reference-defaults > gpurun_out/r05/b_bench_s$sched.json 2>gpurun_out/r05/b_bench_s$sched.err || { echo "bench sched $sched failed"; tail -3 gpurun_out/r05/b_bench_s$sched.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05/b_bench_s$sched.json").read().strip().split("\n")[-1])
ib=d["roofline"].get("issue_bound") or {}
print("sched $sched: value %.1f kernel_ms %.4f valu %.4g lane_util %.3f traffic %s" % (d["value"], d["roofline"]["kernel_ms"], ib.get("valu_insts_per_launch",0), ib.get("lane_util",0), d["roofline"].get("traffic")))
PY
done
