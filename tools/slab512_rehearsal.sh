#!/bin/bash
# One rank's slab of BASELINE configs[4] (512 x 512 x 64 planes of the 512^3 grid) on ONE GPU, its two halo planes
# travelling through RCCL send/recv to itself (ITS_BENCH_FORCE_DIST=2): bench line, then FETCH_SIZE / WRITE_SIZE of the
# in-loop SpMV for traffic / streamed bytes of that shape.   usage: tools/slab512_rehearsal.sh [tag]
TAG=${1:-slab512}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$TAG; mkdir -p gpurun_out/$TAG
export ITS_BENCH_FORCE_DIST=2 ITS_BENCH_NO_PIPELINED=1
timeout -k 10 300 python3 bench.py --global-grid 512 --no-cpu > gpurun_out/$TAG/bench_loopback.json 2> gpurun_out/$TAG/bench.err || { tail -5 gpurun_out/$TAG/bench.err; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$TAG/pmc_$c -- python3 bench.py --global-grid 512 --steps 5 --warmup 2 --no-cpu > gpurun_out/$TAG/pmc_$c.log 2>&1 || { tail -5 gpurun_out/$TAG/pmc_$c.log; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
T = sys.argv[1]
out = open(f'gpurun_out/{T}/pmc_summary_slab512.txt', 'w')
for cname in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/{T}/pmc_{cname}/*/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            if 'its::' in k[0]:
                print('%-12s n=%3d avg=%14.1f KB  %s' % (k[1], len(v), sum(v) / len(v), k[0][:110]), file=out)
out.close()
print(open(f'gpurun_out/{T}/pmc_summary_slab512.txt').read())
PY
cat gpurun_out/$TAG/bench_loopback.json
