#!/bin/bash
# round 4, run o: the shadow walk's stream at high / low priority (VSPG_WF_S2_PRIO), ms per step
mkdir -p gpurun_out/r04
O=gpurun_out/r04
: > $O/o_prio.txt
for rep in 1 2; do for wl in cloud cloud-nvdb cloud-scene; do for prio in default high low; do
  ms=$(env VSPG_WF_S2_PRIO=$prio timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-generic --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "rep$rep $wl $prio $ms" | tee -a $O/o_prio.txt
done; done; done
exit 0
