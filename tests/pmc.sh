#!/bin/bash
# separate --pmc passes (FETCH_SIZE uses 3 TCC slots, WRITE_SIZE 2): one run each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu --spmv-reps 3 > gpurun_out/pmc_$tag.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            print(d.split('/')[-2], k, 'n=%d avg=%.1f' % (len(v), sum(v)/len(v)))
PY
