#!/bin/bash
# round 4: scene-file boundary test, the shadow-query counter through the boundary + cloud tests, A/B of the boundary code's cost on the
# unbounded cloud workloads, the gather-granularity microbenchmark with the L2's request-size counters
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=6 step 400 e_scenefile.log python -m pytest tests/test_host_adapter.py -m gpu -q -x -k "boundaries"
TAILN=6 step 600 e_tests.log python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "boundaries or cloud or grid or nvdb or nds or wavefront or shard"
WORKLOADS="cloud cloud-nvdb cloud-guided" TAILN=14 step 900 e_variants.log scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $O/e_variants_wl.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
$R/scripts/microbench/gather_lines > $R/$O/e_gather_lines.txt 2>&1; echo "gather rc=$?"; cat $R/$O/e_gather_lines.txt
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/gl_$tag
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/gl_$tag -- $R/scripts/microbench/gather_lines > /tmp/gl_$tag.log 2>&1 || { echo "pmc $tag failed"; tail -3 /tmp/gl_$tag.log; }
  f=$(find /tmp/gl_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $R/$O/e_gather_lines.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not k.startswith("k_") and "k_coal" not in k and "k_rec32" not in k and "k_stream" not in k: continue
    print(k, {c: round(sum(v) / len(v) / 67108864.0, 4) for c, v in acc[k].items()}, "(per block; 2^26 blocks per launch; k_stream: per 64 B)")
PY
  fi
done
cat $R/$O/e_gather_lines.txt | tail -30
exit 0
