for cfg in "ITS_SPMV_PANEL_LOG2PW=16" "ITS_SPMV_PANEL_LOG2PW=15" "ITS_SPMV_PANEL_LOG2PW=17" "ITS_SPMV_PANEL_LOG2PW=14"; do
  env $cfg timeout -k 10 300 python bench_configs.py c4 --no-cpu > gpurun_out/c4_tmp.log 2>&1 || exit 1
  python3 -c "
import json
out=[]
for l in open('gpurun_out/c4_tmp.log'):
    d=json.loads(l)
    if d.get('what') in ('spmv_exact','jacobi (correct sweep)') or 'gauss' in str(d.get('what')): out.append('%s %.4f' % (d['what'][:10], d.get('ms') or d.get('ms_per_trip')))
print('$cfg', out)
"
done
