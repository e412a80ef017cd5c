#!/bin/bash
# end-to-end its_solve on a generated 128^3 7-pt Laplacian .mtx: parse + preprocessing/upload + solve, with the
# fscanf reader, the parallel reader and the binary cache
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, time; sys.path.insert(0, '.')
from iterative_solvers_amd import matgen
rp, col, val = matgen.laplace3d_7pt(128)
t = time.time(); matgen.write_mtx('/tmp/lap128.mtx', rp, col, val); print('wrote /tmp/lap128.mtx in %.1f s' % (time.time() - t), flush=True)
PY
E=iterative_solvers_amd/host/its_solve
for mode in "ITS_MTX_READER=fscanf" "ITS_MTX_CACHE=0" "ITS_MTX_CACHE=1" "ITS_MTX_CACHE=1"; do
  echo "== $mode"
  t0=$(date +%s.%N)
  env $mode ITS_TOL=1e-10 timeout -k 10 300 $E /tmp/lap128.mtx -cg -scale none > /tmp/e2e.out 2>&1 || { tail -5 /tmp/e2e.out; exit 1; }
  t1=$(date +%s.%N)
  grep -E "converged|elapsed|Preprocessing time|Harness" /tmp/e2e.out
  python3 -c "print('wall %.2f s' % ($t1 - $t0))"
done
