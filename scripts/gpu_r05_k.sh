#!/bin/bash
# round 5, step k: two ranks of bench.py on the one card (VSPG_BENCH_REHEARSE=1: both on device 0, collectives over gloo -- never a
# benchmark number): the N-rank code path of the rewritten sharding.py (statistics exchange, film all-reduce) on the HIP renderer
mkdir -p gpurun_out/r05; : > gpurun_out/r05/rehearse2.txt
p=29541
for wl in fog fog-guided cloud; do
  p=$((p+1))
  VSPG_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p \
      bench.py --gpus 2 --workload $wl --steps 8 --warmup 2 --train-waves 4 > gpurun_out/r05/rehearse2_$wl.log 2>&1 || { echo "$wl FAILED"; tail -15 gpurun_out/r05/rehearse2_$wl.log; exit 1; }
  tail -1 gpurun_out/r05/rehearse2_$wl.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$wl', 'n_gpus', d['n_gpus'], 'value', round(d['value'],1), d['unit'], 'ms/step', round(d['ms_per_step'],3), 'film_weight_ok', d.get('film_weight_ok'), 'scaling', d['scaling'])" | tee -a gpurun_out/r05/rehearse2.txt
done
