#!/bin/bash
# kernel-trace summary of the C4 multicolour Gauss-Seidel trip, fused vs separate residual pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 1 0; do
  export FUSED=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gsp$f -- python3 tools/gs_probe.py ${1:-10000000} > gpurun_out/gsp$f.log 2>&1 || exit 1
  grep "ms/trip" gpurun_out/gsp$f.log
  cat gpurun_out/gsp$f/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150 | head -8
done
