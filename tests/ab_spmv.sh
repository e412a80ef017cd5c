#!/bin/bash
for cfg in "4 512 1" "4 768 1" "5 512 1" "2 1024 1"; do
  set -- $cfg
  ITS_SPMV_VARIANT=$1 ITS_SPMV_GRID=$2 ITS_SPMV_MAP=$3 timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant $1 grid $2 map $3: cg ms/step %.4f  spmv_dot ms %.4f (%.0f GB/s) exact %.4f vector %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['spmv']['exact']['ms'], d['spmv']['vector']['ms']))"
done
