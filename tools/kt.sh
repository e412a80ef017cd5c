#!/bin/bash
# kernel-trace statistics of one python tool run: tools/kt.sh <out name> <script> [args...]   (prints the top kernels)
set -o pipefail
name=$1; shift
rm -rf gpurun_out/$name; mkdir -p gpurun_out/$name
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python3 "$@" > gpurun_out/$name/log.txt 2>&1 || { tail -5 gpurun_out/$name/log.txt; exit 1; }
python3 - "$name" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-72s calls %6s avg %10.1f us  min %9.1f  max %9.1f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
