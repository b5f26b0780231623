#!/bin/bash
# round 5: four-wide BVH: the triangle / scenario tests, then the terrain timings (profiles/r04_tri_timing.txt is the binary tree's)
mkdir -p gpurun_out/r05
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_reference_scenarios.py -q -x -k "triangle or tri or watertight or reintersect or bad_cases or boundaries or guiding_with_triangles" -m gpu > gpurun_out/r05/g_tests.log 2>&1; rc=$?
grep -c "Memory access fault" gpurun_out/r05/g_tests.log; tail -4 gpurun_out/r05/g_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/tri_timing.py > gpurun_out/r05/g_tri_timing.txt 2>&1 || { tail -5 gpurun_out/r05/g_tri_timing.txt; exit 1; }
cat gpurun_out/r05/g_tri_timing.txt
