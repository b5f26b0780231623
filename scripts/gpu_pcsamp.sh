#!/bin/bash
# PC sampling of the render kernel (diagnostic).  $1 = method (stochastic|host_trap)
set -o pipefail
M=${1:-stochastic}
OUT=$PWD/gpurun_out/pcsamp_$M
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$M" = "stochastic" ]; then UNIT=cycles; INT=1048576; else UNIT=time; INT=1; fi
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $M --pc-sampling-interval $INT --output-format csv json -d $OUT/run -- python3 $REPO/bench.py --steps 40 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
echo "rc=$?"; tail -5 $OUT/log.txt; find $OUT -type f | head -20; du -sh $OUT
