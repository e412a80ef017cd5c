#!/bin/bash
# Produces the per-round evidence under gpurun_out/ (copied into profiles/ afterwards):
#   bench JSON line, rocprofv3 --kernel-trace --stats summary of the same command, and the
#   HBM traffic counters (separate --pmc passes, FETCH_SIZE and WRITE_SIZE cannot share one).
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$R      # (on the GPU box; gpurun MERGES results into the local gpurun_out/: remove the local
                          # gpurun_out/$R before the call as well, or stale traces sit beside the new ones)
mkdir -p gpurun_out/$R
timeout -k 10 400 python3 bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace -- python3 bench.py --no-cpu --no-extra --no-generic > gpurun_out/$R/trace.log 2>&1 || exit 1
export ITS_BENCH_NO_PERMUTED=1      # the counter passes also profile spmv_ellg_kernel on the C2 matrix (spmv.exact_generic)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$R/pmc_$c -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra --spmv-reps 5 > gpurun_out/$R/pmc_$c.log 2>&1 || exit 1
done
python3 - "$R" <<'PY'
import csv, glob, collections, sys
R=sys.argv[1]
out=open(f'gpurun_out/{R}/pmc_summary.txt','w')
import subprocess
print('# kernel_sources_sha256=' + subprocess.check_output(['python3','bench.py','--print-source-hash'], text=True).strip(), file=out)
for cname in ('FETCH_SIZE','WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/{R}/pmc_{cname}/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            print('%-12s n=%3d avg=%14.1f KB  %s' % (k[1], len(v), sum(v)/len(v), k[0][:110]), file=out)
out.close()
print(open(f'gpurun_out/{R}/pmc_summary.txt').read())
PY
cat gpurun_out/$R/trace/*/*kernel_stats.csv | cut -c1-200
cat gpurun_out/$R/bench.json
