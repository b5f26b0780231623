#!/bin/bash
# round 5: both walks of an iteration in one kernel (k_wf_walk): pipeline tests, then same-box A/B against the two-kernel schedule
mkdir -p gpurun_out/r05
rc=0

[ $rc -eq 0 ] || exit $rc
: > gpurun_out/r05/i_ab.txt
for rep in 1 2; do for wl in cloud cloud-nvdb cloud-guided cloud-scene cloud-scene-guided; do for m in 1 0; do
  ms=$(VSPG_WF_MERGED=$m timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 2 --train-waves 4 --no-cpu-baseline --no-generic --no-pmc --no-fast-arith 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "rep$rep $wl merged=$m $ms" | tee -a gpurun_out/r05/i_ab.txt
done; done; done
