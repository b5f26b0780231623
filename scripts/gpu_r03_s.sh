#!/bin/bash
# fused vertex kernel + concurrent walks: parity tests of the grid / NanoVDB pipelines, then same-box A/B against the round's
# earlier library (build/variants/old.so) and against the single-stream order (VSPG_WF_SERIAL=1)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03s
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_nvdb_reader.py -x -q -m gpu -k "grid or nvdb or cloud or wavefront or placed or explosion or heterogeneous or pipeline or config5 or triangle or infinite or light" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
WORKLOADS="cloud cloud-nvdb cloud-guided" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $OUT/
for wl in cloud cloud-nvdb; do
  ms=$(VSPG_WF_SERIAL=1 timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-generic --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "serial $wl $ms" | tee -a $OUT/variants_wl.txt
done
