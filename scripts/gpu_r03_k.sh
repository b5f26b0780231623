#!/bin/bash
mkdir -p gpurun_out/r03k
O=gpurun_out/r03k
( time timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3; tail -c 3000 $O/bench_default.json; tail -3 $O/bench_default.err
timeout -k 10 300 python bench.py --workload cloud --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_cloud.json 2> $O/bench_cloud.err; tail -c 1500 $O/bench_cloud.json
exit 0
