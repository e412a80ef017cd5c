#!/bin/bash
# One multicolour Gauss-Seidel trip on the power-law matrix, launch by launch: start, duration, gap to the next launch
# (rocprofv3 --kernel-trace), and per launch SQ_WAVES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE (a second run, counters only).
# usage: tools/c4_colour_trace.sh [rows]      -> gpurun_out/c4trace/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-10000000}
rm -rf gpurun_out/c4trace; mkdir -p gpurun_out/c4trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c4trace/kt -- python3 tools/gs_probe.py $N > gpurun_out/c4trace/kt.log 2>&1 || { tail -5 gpurun_out/c4trace/kt.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/c4trace/pmc -- python3 tools/gs_probe.py $N > gpurun_out/c4trace/pmc.log 2>&1 || { tail -5 gpurun_out/c4trace/pmc.log; exit 1; }
python3 tools/c4_colour_trace.py gpurun_out/c4trace > gpurun_out/c4trace/summary.txt
cat gpurun_out/c4trace/summary.txt
