#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the exact SpMV for a few tunable settings (one rocprofv3 pass each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1))
  for c in FETCH_SIZE WRITE_SIZE; do
    REPS=4 ROUNDS=1 timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcab_${i}_$c -- python3 tools/spmv_ab.py "$cfg" > gpurun_out/pmcab_${i}_$c.log 2>&1 || exit 1
  done
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys
for i, cfg in enumerate(sys.argv[1:], 1):
    for cname in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in glob.glob(f'gpurun_out/pmcab_{i}_{cname}/*/*counter_collection.csv'):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'spmv' in r['Kernel_Name']:
                    acc[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
            for k, v in sorted(acc.items()):
                print('%-28s %-11s n=%3d avg=%12.1f KB  %s' % (cfg, k[1], len(v), sum(v) / len(v), k[0]))
PY
