#!/bin/bash
# round 4: the full-scene workgroup kernel (triangles, spheres, infinite lights, boundaries over homogeneous media) -- the whole GPU suite, then timings
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=25 step 900 h_tests.log python -m pytest tests -m gpu -q
TAILN=12 step 300 h_tri_timing.txt python scripts/tri_timing.py
VSPG_KERNEL=lane TAILN=12 step 300 h_tri_timing_lane.txt python scripts/tri_timing.py
TAILN=1 step 400 h_bench_default.json python bench.py --no-cpu-baseline
exit 0
