#!/bin/bash
# round 5: same-box A/B of library builds on the terrain scenes (fog + triangles: the FULL workgroup kernel), then the phase profile
mkdir -p gpurun_out/r05
: > gpurun_out/r05/h_tri_variants.txt
for rep in 1 2; do
for lib in "" $(ls build/variants/*.so 2>/dev/null); do
  out=$(VSPG_TRI_ONLY_FOG=1 VSPG_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python scripts/tri_timing.py 2>&1 | grep "^fog" | sed 's/-triangle terrain: k_render_wave_wg2<HomogeneousMedium>//' | tr '\n' ';')
  echo "rep$rep ${lib:-default} $out" | tee -a gpurun_out/r05/h_tri_variants.txt
done; done
timeout -k 10 100 python scripts/prof_sections.py > gpurun_out/r05/prof_wg3_final.txt 2>&1; tail -18 gpurun_out/r05/prof_wg3_final.txt
