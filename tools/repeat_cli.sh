cd $GRAFT_REPO_ROOT
M=$(python3 -c "
import sys; sys.path.insert(0,'tests')
from helpers import mtx_path
print(mtx_path('lap3d_16'))")
for i in $(seq 1 30); do
  ITS_GMRES_RESTART_LEN=100 ITS_TOL=1e-10 timeout -k 5 60 iterative_solvers_amd/host/its_solve $M -gm -scale none > gpurun_out/rep_$i.txt 2>&1 || { echo "run $i failed"; exit 1; }
done
md5sum gpurun_out/rep_*.txt | awk '{print $1}' | sort | uniq -c
for m in 1 2; do ITS_READBACK_SYNC=1 ITS_GMRES_RESTART_LEN=100 ITS_TOL=1e-10 iterative_solvers_amd/host/its_solve $M -gm -scale none > gpurun_out/rep_sync_$m.txt 2>&1; done
md5sum gpurun_out/rep_sync_*.txt gpurun_out/rep_1.txt
