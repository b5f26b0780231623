#!/bin/bash
# round 4, run q: which walk gets the caller's stream (VSPG_WF_SWAP) x the second stream's priority, ms per step; tests under the swap first
mkdir -p gpurun_out/r04
O=gpurun_out/r04
VSPG_WF_SWAP=1 timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "cloud or boundaries or temperature" > $O/q_tests.log 2>&1; echo "q_tests rc=$?"; tail -3 $O/q_tests.log
: > $O/q_swap.txt
for rep in 1 2; do for wl in cloud cloud-nvdb cloud-guided cloud-scene; do for sw in 0 1; do for prio in default high low; do
  ms=$(env VSPG_WF_SWAP=$sw VSPG_WF_S2_PRIO=$prio timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --train-waves 4 --no-cpu-baseline --no-generic --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "rep$rep $wl swap=$sw prio=$prio $ms" | tee -a $O/q_swap.txt
done; done; done; done
exit 0
