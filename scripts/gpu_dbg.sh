#!/bin/bash
mkdir -p gpurun_out
NOTRI=1 timeout -k 10 60 python scripts/dbg_tri_cloud.py > gpurun_out/dbg0.log 2>&1; echo "notri rc=$?"; grep -v "^  File\|^$" gpurun_out/dbg0.log | head -8
timeout -k 10 60 python scripts/dbg_tri_cloud.py > gpurun_out/dbg.log 2>&1
rc=$?; echo rc=$rc; grep -v "^  File\|^$" gpurun_out/dbg.log | head -20; exit $rc
