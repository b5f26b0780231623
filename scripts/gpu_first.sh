#!/bin/bash
# first GPU contact: parity tests, smoke, short bench.  Stops at the first step that times out.
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests -m gpu -q -s > gpurun_out/tests_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/tests_gpu.log
tail -25 gpurun_out/tests_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?
echo "smoke rc=$rc"; tail -5 gpurun_out/smoke.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 16 --warmup 2 > gpurun_out/bench_first.log 2>&1
rc=$?
echo "bench rc=$rc"; tail -5 gpurun_out/bench_first.log
exit 0
