// its_dist.hip — multi-GPU layer (row-block partition, halo exchange, scalar all-reduce over
// RCCL).  PLACEHOLDER for the first single-GPU bring-up: reports "not distributed".
#include "../../include/its_hip.h"
#include "its_kernels.h"
struct its_handle;
int its_dist_active(const its_handle *) { return 0; }
int its_dist_spmv(its_handle *, const its::SpmvArgs &, int) { return ITS_EUNSUPPORTED; }
int its_dist_allreduce(its_handle *, double *, int, int) { return ITS_EUNSUPPORTED; }
void its_dist_destroy(its_handle *) {}
extern "C" int its_comm_unique_id(void *) { return ITS_EUNSUPPORTED; }
extern "C" int its_comm_init(its_handle *, int, int, const void *) { return ITS_EUNSUPPORTED; }
extern "C" int its_upload_crs_dist(its_handle *, int64_t, int64_t, int, int64_t, const int *, const int64_t *,
                                   const double *, int) { return ITS_EUNSUPPORTED; }
