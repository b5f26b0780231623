#!/bin/bash
# workgroup-wavefront kernel vs per-lane kernel: parity tests on both, then bench both.
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -m gpu -q -s > gpurun_out/tests_gpu_wg.log 2>&1
rc=$?
echo "pytest(wg) rc=$rc"; tail -8 gpurun_out/tests_gpu_wg.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
VSPG_KERNEL=wg timeout -k 10 300 python bench.py --steps 16 --warmup 2 > gpurun_out/bench_wg.log 2>&1
rc=$?
echo "bench(wg) rc=$rc"; tail -3 gpurun_out/bench_wg.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
VSPG_KERNEL=lane timeout -k 10 300 python bench.py --steps 16 --warmup 2 > gpurun_out/bench_lane.log 2>&1
rc=$?
echo "bench(lane) rc=$rc"; tail -3 gpurun_out/bench_lane.log
exit 0
