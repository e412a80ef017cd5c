#!/bin/bash
# L2 <-> fabric counters of the exact SpMV: write/read request counts and credit stalls
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
G2="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum"
G3="TCC_CYCLE_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_LATENCY_FIFO_FULL_sum"
G4="TCC_HIT_sum TCC_MISS_sum TCC_IB_STALL_sum TCC_BUBBLE_sum"
i=0
for cfg in "$@"; do
  i=$((i+1)); g=0
  for c in "$G1" "$G2" "$G3" "$G4"; do
    g=$((g+1))
    REPS=4 ROUNDS=1 timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/tcc_${i}_$g -- python3 tools/spmv_ab.py "$cfg" > gpurun_out/tcc_${i}_$g.log 2>&1 || { tail -5 gpurun_out/tcc_${i}_$g.log; exit 1; }
  done
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys
for i, cfg in enumerate(sys.argv[1:], 1):
    for g in (1, 2, 3, 4):
        for f in glob.glob(f'gpurun_out/tcc_{i}_{g}/*/*counter_collection.csv'):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'spmv' in r['Kernel_Name'] and 'false' in r['Kernel_Name'][:70]:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
            for k, v in sorted(acc.items()):
                print('%-24s %-40s n=%3d avg=%16.1f' % (cfg, k, len(v), sum(v) / len(v)))
PY
