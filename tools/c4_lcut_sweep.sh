#!/bin/bash
# C4 (power-law, 10 M rows): rows of more than LCUT entries leave the panel image for the wave-per-row launch.  Where is the cut best?
for cfg in "ITS_SPMV_PANEL_LCUT=128" "ITS_SPMV_PANEL_LCUT=256" "ITS_SPMV_PANEL_LCUT=512" "ITS_SPMV_PANEL_LCUT=1024" "ITS_SPMV_PANEL_LCUT=2048"; do
  env $cfg timeout -k 10 300 python bench_configs.py c4 --no-cpu > gpurun_out/c4_tmp.log 2>&1 || { tail -5 gpurun_out/c4_tmp.log; exit 1; }
  python3 -c "
import json
out=[]
for l in open('gpurun_out/c4_tmp.log'):
    if not l.startswith('{'): continue
    d=json.loads(l)
    if d.get('what') in ('spmv_exact','jacobi (correct sweep)') or 'gauss' in str(d.get('what')): out.append('%s %.4f' % (d['what'][:10], d.get('ms') or d.get('ms_per_trip')))
print('$cfg', out)
"
done
