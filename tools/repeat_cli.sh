#!/bin/bash
# Repeats one its_solve run and counts distinct residual histories (a read-back race would show as an outlier).
# usage: tools/repeat_cli.sh [runs=30] [matrix=lap3d_16] [flag=-gm]
cd $GRAFT_REPO_ROOT
RUNS=${1:-30}; MAT=${2:-lap3d_16}; FLAG=${3:--gm}
M=$(python3 -c "
import sys; sys.path.insert(0,'tests')
from helpers import mtx_path
print(mtx_path('$MAT'))")
mkdir -p gpurun_out/rep; rm -f gpurun_out/rep/*.txt
for i in $(seq 1 $RUNS); do
  ITS_GMRES_RESTART_LEN=100 ITS_TOL=1e-10 timeout -k 5 60 iterative_solvers_amd/host/its_solve $M $FLAG -scale none 2>&1 | grep "_infty = " | md5sum >> gpurun_out/rep/sums.txt || { echo "run $i failed"; exit 1; }
  if [ $((i % 100)) -eq 0 ]; then echo "run $i"; fi
done
sort gpurun_out/rep/sums.txt | uniq -c
