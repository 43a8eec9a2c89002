#!/bin/bash
# round 4, after the lazy narrow copy: every measurement of tools/r4_final.sh again with the final binary, the
# static-matrix experiment (tools/r4_static_matrix.py) and the stress tools
mkdir -p gpurun_out/r4 gpurun_out/r4l
bash tools/r4_final.sh > gpurun_out/r4/final2.log 2>&1
tail -60 gpurun_out/r4/final2.log
timeout 300 python tools/r4_static_matrix.py 16384 > gpurun_out/r4/static_matrix_n16384.txt 2>&1
cat gpurun_out/r4/static_matrix_n16384.txt
bash tools/r4_stress_final.sh
