#!/bin/bash
mkdir -p gpurun_out
bash scripts/gpu_wf.sh || exit $?
bash scripts/gpu_wf_prof.sh b cloud || exit $?
bash scripts/gpu_wf_prof.sh c cloud-nvdb
