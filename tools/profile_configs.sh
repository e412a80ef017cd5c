#!/bin/bash
# Evidence for the configs bench.py does not headline (C3 GMRES(50) 27-pt 128^3, C4 power-law 10 M rows):
#   rocprofv3 --kernel-trace --stats summary + PMC passes (one counter group per run, program directly after --)
#   of `python3 bench_configs.py <cfg> --no-cpu`, summarised per kernel into gpurun_out/<tag>/ (copy to profiles/).
# usage: tools/profile_configs.sh <tag> <c3|c4> [--small]
TAG=$1; CFG=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$TAG; mkdir -p gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench_configs.py $CFG --no-cpu "$@" > gpurun_out/$TAG/trace.log 2>&1 || { tail -5 gpurun_out/$TAG/trace.log; exit 1; }
G1="FETCH_SIZE"
G2="WRITE_SIZE"
G3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
G4="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum"
G5="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
g=0
for c in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  g=$((g+1))
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$TAG/pmc_$g -- python3 bench_configs.py $CFG --no-cpu "$@" > gpurun_out/$TAG/pmc_$g.log 2>&1 || { tail -5 gpurun_out/$TAG/pmc_$g.log; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys, subprocess
T = sys.argv[1]
out = open(f'gpurun_out/{T}/pmc_summary.txt', 'w')
print('# kernel_sources_sha256=' + subprocess.check_output(['python3', 'bench.py', '--print-source-hash'], text=True).strip(), file=out)
print('# per-launch averages; FETCH_SIZE / WRITE_SIZE in KB (gfx950: FETCH_SIZE counts 64 B per 128-B request of wide reads)', file=out)
for g in range(1, 6):
    for f in glob.glob(f'gpurun_out/{T}/pmc_{g}/*/*counter_collection.csv'):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(acc.items()):
            if k[0].startswith('__amd') or 'its::' not in k[0]:
                continue
            print('%-34s n=%4d avg=%16.1f  %s' % (k[1], len(v), sum(v) / len(v), k[0][:100]), file=out)
out.close()
PY
cp gpurun_out/$TAG/trace/*/*kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv
cat gpurun_out/$TAG/trace.log | grep '^{' > gpurun_out/$TAG/lines.jsonl
echo done
