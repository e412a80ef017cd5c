#!/bin/bash
# SQ-side counters of the exact SpMV (one rocprofv3 pass per counter group), config strings as in spmv_ab.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM"
G2="SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES"
G3="GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum"
i=0
for cfg in "$@"; do
  i=$((i+1)); g=0
  for c in "$G1" "$G2" "$G3"; do
    g=$((g+1))
    REPS=4 ROUNDS=1 timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/sq_${i}_$g -- python3 tools/spmv_ab.py "$cfg" > gpurun_out/sq_${i}_$g.log 2>&1 || { tail -5 gpurun_out/sq_${i}_$g.log; exit 1; }
  done
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys
for i, cfg in enumerate(sys.argv[1:], 1):
    for g in (1, 2, 3):
        for f in glob.glob(f'gpurun_out/sq_{i}_{g}/*/*counter_collection.csv'):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'spmv' in r['Kernel_Name'] and 'true' not in r['Kernel_Name'][:60]:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
            for k, v in sorted(acc.items()):
                print('%-24s %-34s n=%3d avg=%16.1f' % (cfg, k, len(v), sum(v) / len(v)))
PY
