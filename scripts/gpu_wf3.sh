#!/bin/bash
mkdir -p gpurun_out
bash scripts/gpu_wf.sh || exit $?
bash scripts/gpu_wf_pmc.sh cloud cloud || exit $?
bash scripts/gpu_wf_pmc.sh nvdb cloud-nvdb
