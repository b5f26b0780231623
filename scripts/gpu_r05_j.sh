#!/bin/bash
# round 5, step j: the pipeline's majorant reads as LDS instructions (GridMediumT's MAJLDS) -- parity subset, then the same-box A/B
# against build/variants/majgen.so (-DVSPG_WF_MAJ_LDS=0 on vspg_wf_grid.hip: flat loads through the generic pointer, as before)
mkdir -p gpurun_out/r05
timeout -k 10 700 python -m pytest tests -m gpu -q -x -k "grid or cloud or merged or boundar or blackbody or temperature or pipeline" > gpurun_out/r05/tests_j.log 2>&1 || { tail -20 gpurun_out/r05/tests_j.log; exit 1; }
tail -2 gpurun_out/r05/tests_j.log
WORKLOADS="cloud cloud-scene cloud-guided cloud-scene-guided" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt gpurun_out/r05/majlds_ab.txt
