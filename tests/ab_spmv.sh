#!/bin/bash
for abl in 0 1 2 3; do
  ITS_SPMV_ABLATE=$abl timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ablate $abl: spmv_dot ms %.4f (%.0f GB/s) exact %.4f' % (d['roofline']['kernel_ms'], d['roofline']['achieved'], d['spmv']['exact']['ms']))"
done
