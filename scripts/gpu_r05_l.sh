#!/bin/bash
# round 5, step l: k_wf_vertex regroups the links of a no-walk chain over the workgroup's lanes (boundary scenes) -- parity subset,
# the same-box A/B against build/variants/noregroup.so (-DVSPG_WF_REGROUP=0: the lane that began a chain runs it), the 2-rank rehearsal
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "boundar or merged or cloud or scene or pipeline or grid" > gpurun_out/r05/tests_l.log 2>&1 || { tail -30 gpurun_out/r05/tests_l.log; exit 1; }
tail -2 gpurun_out/r05/tests_l.log
WORKLOADS="cloud-scene cloud-scene-nvdb cloud" bash scripts/gpu_variants_wl.sh && cp gpurun_out/variants_wl.txt gpurun_out/r05/regroup_ab.txt && bash scripts/gpu_r05_k.sh
