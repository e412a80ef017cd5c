#!/bin/bash
# fabric-side latency of reads with and without a mixed-in write stream (tools/streamlab kernels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum --output-format csv -d gpurun_out/lab_pmc -- ./tools/streamlab 256 > gpurun_out/lab_pmc.log 2>&1 || { tail -5 gpurun_out/lab_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/lab_pmc/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    rd, rl, wr, wl = (m.get(c, 0.0) for c in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_LEVEL_sum', 'TCC_EA0_WRREQ_sum', 'TCC_EA0_WRREQ_LEVEL_sum'))
    print('%-70s rd %9.0f lat %7.1f cyc | wr %9.0f lat %7.1f cyc' % (k, rd, rl / rd if rd else 0, wr, wl / wr if wr else 0))
PY
