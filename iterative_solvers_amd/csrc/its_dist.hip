// its_dist.hip — multi-GPU layer of libits_hip.so: one process per GPU, matrix row-block
// partitioned.  Per SpMV (default, "dist_overlap" 0): pack the x entries the neighbours need
// (skipped when every send list is one ascending run of rows: sent straight out of the vector), exchange them with
// grouped RCCL send/recv over xGMI ON THE COMPUTE STREAM, then ONE product over the ordinary row-block plan.
// Exchange-first is the default because it measured faster than hiding the exchange: beside a persistent SpMV
// grid the RCCL kernel's workgroups wait 100-330 us for CUs.  The overlapped form (exchange on a
// high-priority side stream beside the interior row blocks, boundary blocks after the halo event) is kept behind
// option "dist_overlap" 1.  Per dot/norm: one RCCL all-reduce of the ITS_NPART partials on the compute stream —
// no host synchronisation anywhere in a trip.
//
// The reference has no counterpart (single process; "TODO: MPI", main.cpp:94).  The partition is
// planned on the host (iterative_solvers_amd/distributed.py) and arrives through its_dist_setup.
//
// Two transports:
//   RCCL  (its_comm_init)       production.
//   HOST  (its_comm_init_host)  test transport: halo data and scalars are staged through host
//                               callbacks, so several ranks can share ONE GPU and the whole
//                               pack / split-SpMV / unpack / reduce sequence runs on hardware
//                               where only a single device is available.
#include "its_handle.h"

#include <rccl/rccl.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

static int env_seconds(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && atoi(e) > 0 ? atoi(e) : dflt;
}

struct DistPlan {
    int rank = 0, world = 1;
    int transport = 0;                       // 0 none, 1 RCCL, 2 host callbacks
    ncclComm_t comm_halo = nullptr, comm_red = nullptr;
    hipStream_t cstream = nullptr;
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    its_exchange_cb xcb = nullptr;
    its_allreduce_cb rcb = nullptr;
    void *user = nullptr;
    // exchange plan
    bool plan_ready = false;
    int n_nbr = 0, n_send = 0, n_halo = 0;
    std::vector<int> nbr, send_off, recv_off;
    // when every neighbour's send list is one ascending run of local rows (the planes of a slab), entries are sent
    // straight out of the vector: send_first[k] = first row of the run; empty = pack through send_buf
    std::vector<int> send_first;
    int *send_idx = nullptr;                 // device
    double *send_buf = nullptr;              // device
    double *host_send = nullptr, *host_recv = nullptr, *host_sc = nullptr;   // pinned (HOST transport)
    // multicolour sweeps across ranks (its_dist_set_colors): local rows are ordered colour by colour, and so are
    // every neighbour's send list and halo range; colour c of neighbour k is send_coff[k*(n_colors+1) + c .. + c+1)
    // of send_idx and recv_coff[...] of the halo
    int n_colors = 0;
    std::vector<int> color_ptr, send_coff, recv_coff;
    // row blocks without / with halo columns
    int4 *desc_int = nullptr, *desc_bnd = nullptr;
    int nb_int = 0, nb_bnd = 0;
    // the same split of the larger-block plan (CsrDev::blk_desc_big), for launches that take it
    int4 *desc_int_big = nullptr, *desc_bnd_big = nullptr;
    int nb_int_big = 0, nb_bnd_big = 0;
};

#define NCCLCHK(h, call)                                                                   \
    do {                                                                                    \
        ncclResult_t r_ = (call);                                                           \
        if (r_ != ncclSuccess) {                                                            \
            (h)->err = std::string(#call) + ": " + ncclGetErrorString(r_);                  \
            return ITS_ECOMM;                                                               \
        }                                                                                   \
    } while (0)

__global__ __launch_bounds__(ITS_BLOCK) void pack_kernel(const double *__restrict__ x, const int *__restrict__ idx,
                                                         double *__restrict__ out, int n)
{
    const int i = blockIdx.x * ITS_BLOCK + threadIdx.x;
    if (i < n) out[i] = x[idx[i]];
}

int its_dist_active(const its_handle *h) { return h->dist && h->dist->transport > 0 && h->dist->plan_ready; }

void its_dist_destroy(its_handle *h)
{
    DistPlan *d = h->dist;
    if (!d) return;
    if (d->comm_halo) ncclCommDestroy(d->comm_halo);
    if (d->comm_red) ncclCommDestroy(d->comm_red);
    if (d->cstream) (void)hipStreamDestroy(d->cstream);
    if (d->ev_pack) (void)hipEventDestroy(d->ev_pack);
    if (d->ev_halo) (void)hipEventDestroy(d->ev_halo);
    dfree(d->send_idx); dfree(d->send_buf); dfree(d->desc_int); dfree(d->desc_bnd);
    dfree(d->desc_int_big); dfree(d->desc_bnd_big);
    if (d->host_send) (void)hipHostFree(d->host_send);
    if (d->host_recv) (void)hipHostFree(d->host_recv);
    if (d->host_sc) (void)hipHostFree(d->host_sc);
    delete d;
    h->dist = nullptr;
}

static int ensure_plan(its_handle *h)
{
    if (!h->dist) {
        h->dist = new DistPlan;
        HIPCHK(h, hipSetDevice(h->device));
        // the halo stream outranks the compute stream: pack finishes, then the exchange kernel and the interior
        // SpMV become runnable together, and the (tiny) exchange must get its CUs first or a persistent SpMV
        // grid would hold it back until the interior product is done
        int prio_least = 0, prio_greatest = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIPCHK(h, hipStreamCreateWithPriority(&h->dist->cstream, hipStreamNonBlocking, prio_greatest));
        HIPCHK(h, hipEventCreateWithFlags(&h->dist->ev_pack, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->dist->ev_halo, hipEventDisableTiming));
        HIPCHK(h, hipHostMalloc((void **)&h->dist->host_sc, 2 * ITS_NPART * sizeof(double), hipHostMallocDefault));
    }
    return ITS_OK;
}

extern "C" int its_comm_unique_id(void *id128)
{
    if (!id128) return ITS_EINVAL;
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return ITS_ECOMM;
    memcpy(id128, &id, sizeof id);
    return ITS_OK;
}

extern "C" int its_comm_init(its_handle *h, int rank, int world, const void *ids256)
{
    if (!h || !ids256 || world < 1 || rank < 0 || rank >= world) return ITS_EINVAL;
    int rc = ensure_plan(h);
    if (rc) return rc;
    DistPlan *d = h->dist;
    if (d->transport < 0) FAIL(h, ITS_ESTATE, "an earlier its_comm_init on this handle timed out: the handle cannot join a row-partitioned run (start a fresh process)");
    if (d->transport != 0) FAIL(h, ITS_ESTATE, "communicator already initialised");
    ncclUniqueId id0, id1;
    memcpy(&id0, ids256, 128);
    memcpy(&id1, (const char *)ids256 + 128, 128);
    // two communicators: halo send/recv run on the side stream, all-reduces on the compute stream;
    // operations of ONE communicator must not be in flight on two streams at once
    // ncclCommInitRank blocks until every rank has joined; a rank that never arrives (crashed, wrong WORLD_SIZE,
    // two ranks on one device) would hang the others for good.  The blocking calls therefore run on a helper
    // thread (bound to this handle's device) that the caller waits on for ITS_COMM_TIMEOUT_S (default 180 s).
    HIPCHK(h, hipSetDevice(h->device));
    int dev_count = 0;
    HIPCHK(h, hipGetDeviceCount(&dev_count));
    const int timeout_s = std::max(1, env_seconds("ITS_COMM_TIMEOUT_S", 180));
    struct InitState { std::mutex mu; std::condition_variable cv; bool done = false, abandoned = false;
                       ncclResult_t r0 = ncclSuccess, r1 = ncclSuccess; ncclComm_t c0 = nullptr, c1 = nullptr; };
    auto st = std::make_shared<InitState>();
    const int device = h->device;
    std::thread([st, id0, id1, world, rank, device]() {
        ncclResult_t r0 = ncclSystemError, r1 = ncclSystemError;
        ncclComm_t c0 = nullptr, c1 = nullptr;
        if (hipSetDevice(device) == hipSuccess) {
            r0 = ncclCommInitRank(&c0, world, id0, rank);
            r1 = r0 == ncclSuccess ? ncclCommInitRank(&c1, world, id1, rank) : r0;
        }
        bool abandoned;
        {
            std::lock_guard<std::mutex> lk(st->mu);
            abandoned = st->abandoned;
            if (!abandoned) { st->r0 = r0; st->r1 = r1; st->c0 = c0; st->c1 = c1; }
            st->done = true;
            st->cv.notify_all();
        }
        // the waiter gave up (time-out) before the peers arrived: nobody will ever use or free these communicators
        if (abandoned) { if (c0) (void)ncclCommAbort(c0); if (c1) (void)ncclCommAbort(c1); }
    }).detach();
    {
        std::unique_lock<std::mutex> lk(st->mu);
        if (!st->cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return st->done; })) {
            char msg[512];
            snprintf(msg, sizeof msg, "ncclCommInitRank: rank %d of %d on device %d (of %d visible) did not complete within %d s "
                     "(a peer rank missing, or two ranks on one GPU?); this handle cannot join a row-partitioned run any more: exit "
                     "the rank (non-zero) and start a fresh process", rank, world, device, dev_count, timeout_s);
            st->abandoned = true;                  // the helper thread aborts whatever it obtains later
            d->transport = -1;                     // a second its_comm_init on this handle is refused ("already initialised")
            FAIL(h, ITS_ECOMM, msg);
        }
        if (st->r0 != ncclSuccess) FAIL(h, ITS_ECOMM, std::string("ncclCommInitRank (halo): ") + ncclGetErrorString(st->r0));
        if (st->r1 != ncclSuccess) FAIL(h, ITS_ECOMM, std::string("ncclCommInitRank (reductions): ") + ncclGetErrorString(st->r1));
        d->comm_halo = st->c0; d->comm_red = st->c1;
    }
    d->rank = rank; d->world = world; d->transport = 1;
    return ITS_OK;
}

extern "C" int its_comm_init_host(its_handle *h, int rank, int world, its_exchange_cb xcb,
                                  its_allreduce_cb rcb, void *user)
{
    if (!h || !xcb || !rcb || world < 1 || rank < 0 || rank >= world) return ITS_EINVAL;
    int rc = ensure_plan(h);
    if (rc) return rc;
    DistPlan *d = h->dist;
    if (d->transport != 0) FAIL(h, ITS_ESTATE, "communicator already initialised");
    d->xcb = xcb; d->rcb = rcb; d->user = user;
    d->rank = rank; d->world = world; d->transport = 2;
    return ITS_OK;
}

extern "C" int its_dist_setup(its_handle *h, int n_nbr, const int *nbr, const int *send_off,
                              const int *send_idx, const int *recv_off)
{
    if (!h || n_nbr < 0 || (n_nbr > 0 && (!nbr || !send_off || !recv_off))) return ITS_EINVAL;
    if (n_nbr > 0 && send_off[n_nbr] > 0 && !send_idx) FAIL(h, ITS_EINVAL, "send_idx is null but send_off lists entries to send");
    if (!h->dist || h->dist->transport <= 0) FAIL(h, ITS_ESTATE, "its_comm_init first (and it must have succeeded)");
    if (!h->have[ITS_MAT_A]) FAIL(h, ITS_ESTATE, "upload the local matrix first");
    HIPCHK(h, hipSetDevice(h->device));
    DistPlan *d = h->dist;
    const CsrDev &A = h->mat[ITS_MAT_A];
    const int n_loc = A.n_rows, n_halo = A.n_cols - A.n_rows;
    if (n_halo < 0) FAIL(h, ITS_EINVAL, "local matrix must have n_cols >= n_rows (owned + halo columns)");
    d->n_nbr = n_nbr; d->n_halo = n_halo;
    d->nbr.assign(nbr, nbr + n_nbr);
    d->send_off.assign(1, 0); d->recv_off.assign(1, 0);
    if (n_nbr) { d->send_off.assign(send_off, send_off + n_nbr + 1); d->recv_off.assign(recv_off, recv_off + n_nbr + 1); }
    d->n_send = d->send_off.back();
    // validate what the pack kernel and the receives assume
    if (d->recv_off.back() != n_halo) FAIL(h, ITS_EINVAL, "recv_off must cover the halo exactly");
    for (int k = 0; k < n_nbr; ++k) {
        // (a rank is its own neighbour only in a one-rank world: a periodic problem whose wrap-around entries
        // travel through the same send/recv path — how the RCCL exchange is exercised on a single GPU)
        if (nbr[k] < 0 || nbr[k] >= d->world || (nbr[k] == d->rank && d->world > 1)) FAIL(h, ITS_EINVAL, "bad neighbour rank");
        if (d->send_off[k + 1] < d->send_off[k] || d->recv_off[k + 1] < d->recv_off[k]) FAIL(h, ITS_EINVAL, "offsets not monotone");
    }
    for (int i = 0; i < d->n_send; ++i)
        if (send_idx[i] < 0 || send_idx[i] >= n_loc) FAIL(h, ITS_EINVAL, "send index outside the owned rows");
    d->send_first.clear();
    {
        bool runs = n_nbr > 0;
        for (int k = 0; k < n_nbr && runs; ++k)
            for (int i = d->send_off[k] + 1; i < d->send_off[k + 1] && runs; ++i) runs = send_idx[i] == send_idx[i - 1] + 1;
        if (runs)
            for (int k = 0; k < n_nbr; ++k) d->send_first.push_back(d->send_off[k + 1] > d->send_off[k] ? send_idx[d->send_off[k]] : 0);
    }
    dfree(d->send_idx); dfree(d->send_buf);
    if (dalloc(h, &d->send_idx, d->n_send) || dalloc(h, &d->send_buf, d->n_send)) return ITS_EHIP;
    if (d->n_send) HIPCHK(h, hipMemcpy(d->send_idx, send_idx, (size_t)d->n_send * sizeof(int), hipMemcpyHostToDevice));
    if (d->transport == 2) {
        if (d->host_send) (void)hipHostFree(d->host_send);
        if (d->host_recv) (void)hipHostFree(d->host_recv);
        HIPCHK(h, hipHostMalloc((void **)&d->host_send, std::max(d->n_send, 1) * sizeof(double), hipHostMallocDefault));
        HIPCHK(h, hipHostMalloc((void **)&d->host_recv, std::max(n_halo, 1) * sizeof(double), hipHostMallocDefault));
    }
    // split the row blocks of the stream plan(s): a block is "boundary" if any of its rows has a
    // halo column (>= n_loc); interior blocks can be multiplied while the halo is in flight
    std::vector<int> col((size_t)A.nnz);
    if (A.nnz) HIPCHK(h, hipMemcpy(col.data(), A.col, (size_t)A.nnz * sizeof(int), hipMemcpyDeviceToHost));
    auto split_plan = [&](const int4 *dev_desc, int nb, int4 **out_int, int *n_int, int4 **out_bnd, int *n_bnd) -> int {
        std::vector<int4> desc(std::max(nb, 1));
        if (nb) HIPCHK(h, hipMemcpy(desc.data(), dev_desc, (size_t)nb * sizeof(int4), hipMemcpyDeviceToHost));
        std::vector<int4> di, db;
        for (int b = 0; b < nb; ++b) {
            bool bnd = false;
            for (int k = desc[b].z; k < desc[b].w && !bnd; ++k) bnd = col[k] >= n_loc;
            (bnd ? db : di).push_back(desc[b]);
        }
        *n_int = (int)di.size(); *n_bnd = (int)db.size();
        dfree(*out_int); dfree(*out_bnd);
        if (dalloc(h, out_int, di.size()) || dalloc(h, out_bnd, db.size())) return ITS_EHIP;
        if (!di.empty()) HIPCHK(h, hipMemcpy(*out_int, di.data(), di.size() * sizeof(int4), hipMemcpyHostToDevice));
        if (!db.empty()) HIPCHK(h, hipMemcpy(*out_bnd, db.data(), db.size() * sizeof(int4), hipMemcpyHostToDevice));
        return ITS_OK;
    };
    int rcs = split_plan(A.blk_desc, A.n_blocks, &d->desc_int, &d->nb_int, &d->desc_bnd, &d->nb_bnd);
    if (rcs) return rcs;
    d->nb_int_big = d->nb_bnd_big = 0;
    if (A.blk_desc_big && (rcs = split_plan(A.blk_desc_big, A.n_blocks_big, &d->desc_int_big, &d->nb_int_big, &d->desc_bnd_big, &d->nb_bnd_big)))
        return rcs;
    d->plan_ready = true;
    h->setup_done = false; h->residual_done = false;
    return ITS_OK;
}

// Colour structure of a partition whose local rows are ordered colour by colour (global multicolour ordering,
// SURVEY.md §8e "GS across GPUs: colours are global"): color_ptr[n_colors + 1] cuts the owned rows, and for every
// neighbour k the colour cuts of its send list and of its halo range (absolute offsets into send_idx / the halo,
// n_nbr x (n_colors + 1) each, row-major).  After this call triangular sweeps run colour by colour with one
// exchange of the colour's boundary values after each colour.
extern "C" int its_dist_set_colors(its_handle *h, int n_colors, const int *color_ptr, const int *send_color_off,
                                   const int *recv_color_off)
{
    if (!h || n_colors < 1 || !color_ptr) return ITS_EINVAL;
    if (!h->dist || !h->dist->plan_ready) FAIL(h, ITS_ESTATE, "its_dist_setup first");
    DistPlan *d = h->dist;
    if (d->n_nbr > 0 && (!send_color_off || !recv_color_off)) return ITS_EINVAL;
    const int n_loc = h->mat[ITS_MAT_A].n_rows;
    if (color_ptr[0] != 0 || color_ptr[n_colors] != n_loc) FAIL(h, ITS_EINVAL, "color_ptr must cut [0, owned rows)");
    for (int c = 0; c < n_colors; ++c) if (color_ptr[c + 1] < color_ptr[c]) FAIL(h, ITS_EINVAL, "color_ptr not monotone");
    const int w = n_colors + 1;
    for (int k = 0; k < d->n_nbr; ++k) {
        if (send_color_off[k * w] != d->send_off[k] || send_color_off[k * w + n_colors] != d->send_off[k + 1] ||
            recv_color_off[k * w] != d->recv_off[k] || recv_color_off[k * w + n_colors] != d->recv_off[k + 1])
            FAIL(h, ITS_EINVAL, "colour offsets must cut each neighbour's send list / halo range exactly");
        for (int c = 0; c < n_colors; ++c)
            if (send_color_off[k * w + c + 1] < send_color_off[k * w + c] || recv_color_off[k * w + c + 1] < recv_color_off[k * w + c])
                FAIL(h, ITS_EINVAL, "colour offsets not monotone");
    }
    d->n_colors = n_colors;
    d->color_ptr.assign(color_ptr, color_ptr + w);
    d->send_coff.assign(send_color_off, send_color_off + (size_t)d->n_nbr * w);
    d->recv_coff.assign(recv_color_off, recv_color_off + (size_t)d->n_nbr * w);
    h->setup_done = false;
    return ITS_OK;
}

int its_dist_n_colors(const its_handle *h) { return h->dist ? h->dist->n_colors : 0; }
const int *its_dist_color_ptr(const its_handle *h) { return h->dist && h->dist->n_colors ? h->dist->color_ptr.data() : nullptr; }

// Exchange the boundary values of ONE colour of the vector x (owned entries followed by the halo), or of all
// colours (color < 0), on the compute stream: what a colour step of a triangular sweep publishes to the neighbours.
int its_dist_exchange(its_handle *h, double *x, int color)
{
    DistPlan *d = h->dist;
    if (!d || !d->plan_ready) FAIL(h, ITS_ESTATE, "no exchange plan");
    if (color >= d->n_colors) FAIL(h, ITS_EINVAL, "colour out of range");
    const int n_loc = h->mat[ITS_MAT_A].n_rows;
    const int w = d->n_colors + 1;
    std::vector<int> s0(d->n_nbr), s1(d->n_nbr), r0(d->n_nbr), r1(d->n_nbr);
    int tot_s = 0, tot_r = 0;
    for (int k = 0; k < d->n_nbr; ++k) {
        if (color < 0) { s0[k] = d->send_off[k]; s1[k] = d->send_off[k + 1]; r0[k] = d->recv_off[k]; r1[k] = d->recv_off[k + 1]; }
        else { s0[k] = d->send_coff[k * w + color]; s1[k] = d->send_coff[k * w + color + 1]; r0[k] = d->recv_coff[k * w + color]; r1[k] = d->recv_coff[k * w + color + 1]; }
        tot_s += s1[k] - s0[k]; tot_r += r1[k] - r0[k];
    }
    for (int k = 0; k < d->n_nbr; ++k) {
        const int cnt = s1[k] - s0[k];
        if (cnt > 0)
            hipLaunchKernelGGL(pack_kernel, dim3((cnt + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, h->stream,
                               x, d->send_idx + s0[k], d->send_buf + s0[k], cnt);
    }
    HIPCHK(h, hipGetLastError());
    if (d->transport == 1) {
        if (tot_s + tot_r == 0) return ITS_OK;
        NCCLCHK(h, ncclGroupStart());
        for (int k = 0; k < d->n_nbr; ++k) {
            if (s1[k] > s0[k]) NCCLCHK(h, ncclSend(d->send_buf + s0[k], s1[k] - s0[k], ncclDouble, d->nbr[k], d->comm_halo, h->stream));
            if (r1[k] > r0[k]) NCCLCHK(h, ncclRecv(x + n_loc + r0[k], r1[k] - r0[k], ncclDouble, d->nbr[k], d->comm_halo, h->stream));
        }
        NCCLCHK(h, ncclGroupEnd());
        return ITS_OK;
    }
    // HOST transport: compact the segments, hand them to the callback (every rank calls it, also with nothing to
    // send: the callback is a collective), scatter what came back
    std::vector<int> so(d->n_nbr + 1, 0), ro(d->n_nbr + 1, 0);
    for (int k = 0; k < d->n_nbr; ++k) {
        so[k + 1] = so[k] + (s1[k] - s0[k]); ro[k + 1] = ro[k] + (r1[k] - r0[k]);
        if (s1[k] > s0[k])
            HIPCHK(h, hipMemcpyAsync(d->host_send + so[k], d->send_buf + s0[k], (size_t)(s1[k] - s0[k]) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (d->xcb(d->user, d->host_send, so.data(), d->host_recv, ro.data(), d->nbr.data(), d->n_nbr) != 0)
        FAIL(h, ITS_ECOMM, "host exchange callback failed");
    for (int k = 0; k < d->n_nbr; ++k)
        if (r1[k] > r0[k])
            HIPCHK(h, hipMemcpyAsync(x + n_loc + r0[k], d->host_recv + ro[k], (size_t)(r1[k] - r0[k]) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));       // host_recv is reused by the next exchange
    return ITS_OK;
}

// the formats that are launched over the whole local matrix (no interior / boundary split)
static void launch_spmv_reordered(its_handle *h, const CsrDev &A, const SpmvArgs &a, int epi)
{
    if (h->fmt == ITS_FMT_CSR_VECTOR) launch_spmv_vector(A, a, epi, h->stream);
    else if (h->fmt == ITS_FMT_MIXED) launch_spmv_mixed(A, a, epi, h->stream);
    else launch_spmv_sell(A, a, epi, h->stream);
}

// halo exchange of the vector a.x (length n_loc + n_halo) overlapped with the interior SpMV
int its_dist_spmv(its_handle *h, const SpmvArgs &a, int epi)
{
    DistPlan *d = h->dist;
    const CsrDev &A = h->mat[ITS_MAT_A];
    double *x = const_cast<double *>(a.x);          // solver vectors are ours; the halo tail is scratch
    const int n_loc = A.n_rows;
    const bool direct_send = d->transport == 1 && !d->send_first.empty();
    if (d->n_send && !direct_send) {
        hipLaunchKernelGGL(pack_kernel, dim3((d->n_send + ITS_BLOCK - 1) / ITS_BLOCK), dim3(ITS_BLOCK), 0, h->stream,
                           a.x, d->send_idx, d->send_buf, d->n_send);
        HIPCHK(h, hipGetLastError());
    }
    const bool split = h->fmt == ITS_FMT_CSR_EXACT || epi >= EPI_JACOBI_SHIPPED;
    const bool big = split && spmv_exact_takes_big_plan(A, a, epi);
    const int4 *desc_int = big ? d->desc_int_big : d->desc_int, *desc_bnd = big ? d->desc_bnd_big : d->desc_bnd;
    const int nb_int = big ? d->nb_int_big : d->nb_int, nb_bnd = big ? d->nb_bnd_big : d->nb_bnd;
    if (d->transport == 1 && !h->dist_overlap) {
        // Default: exchange first, on the compute stream, then ONE product over the ordinary plan.  Measured with
        // the one-GPU loopback (bench.py, ITS_BENCH_FORCE_DIST=2): run beside the persistent SpMV grid, the RCCL
        // send/recv kernel's workgroups spin for 100-330 us waiting for CUs and for each other and slow the
        // product down to 415 us (trip 0.637 ms); run alone it takes 6-20 us (trip 0.53 ms).  Hiding a 20 us
        // exchange is not worth a tenth of that.
        if (d->n_nbr) {
            NCCLCHK(h, ncclGroupStart());
            for (int k = 0; k < d->n_nbr; ++k) {
                const int ns = d->send_off[k + 1] - d->send_off[k], nr = d->recv_off[k + 1] - d->recv_off[k];
                if (ns) NCCLCHK(h, ncclSend(direct_send ? a.x + d->send_first[k] : d->send_buf + d->send_off[k], ns, ncclDouble, d->nbr[k], d->comm_halo, h->stream));
                if (nr) NCCLCHK(h, ncclRecv(x + n_loc + d->recv_off[k], nr, ncclDouble, d->nbr[k], d->comm_halo, h->stream));
            }
            NCCLCHK(h, ncclGroupEnd());
        }
        if (split) launch_spmv_exact(A, a, epi, h->stream);
        else launch_spmv_reordered(h, A, a, epi);
    } else if (d->transport == 1) {
        if (d->n_nbr) {
            HIPCHK(h, hipEventRecord(d->ev_pack, h->stream));
            HIPCHK(h, hipStreamWaitEvent(d->cstream, d->ev_pack, 0));
            NCCLCHK(h, ncclGroupStart());
            for (int k = 0; k < d->n_nbr; ++k) {
                const int ns = d->send_off[k + 1] - d->send_off[k], nr = d->recv_off[k + 1] - d->recv_off[k];
                if (ns) NCCLCHK(h, ncclSend(direct_send ? a.x + d->send_first[k] : d->send_buf + d->send_off[k], ns, ncclDouble, d->nbr[k], d->comm_halo, d->cstream));
                if (nr) NCCLCHK(h, ncclRecv(x + n_loc + d->recv_off[k], nr, ncclDouble, d->nbr[k], d->comm_halo, d->cstream));
            }
            NCCLCHK(h, ncclGroupEnd());
            HIPCHK(h, hipEventRecord(d->ev_halo, d->cstream));
        }
        if (split) {
            launch_spmv_exact_desc(A, desc_int, nb_int, a, epi, 0, h->stream);
            if (d->n_nbr) HIPCHK(h, hipStreamWaitEvent(h->stream, d->ev_halo, 0));
            if (nb_bnd) launch_spmv_exact_desc(A, desc_bnd, nb_bnd, a, epi, 1, h->stream);
        } else {
            if (d->n_nbr) HIPCHK(h, hipStreamWaitEvent(h->stream, d->ev_halo, 0));
            launch_spmv_reordered(h, A, a, epi);
        }
    } else {
        // HOST transport (tests): stage through the callbacks; interior blocks are still launched
        // first so the launch sequence matches production
        if (split) launch_spmv_exact_desc(A, desc_int, nb_int, a, epi, 0, h->stream);
        if (d->n_send) HIPCHK(h, hipMemcpyAsync(d->host_send, d->send_buf, (size_t)d->n_send * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (d->xcb(d->user, d->host_send, d->send_off.data(), d->host_recv, d->recv_off.data(), d->nbr.data(), d->n_nbr) != 0)
            FAIL(h, ITS_ECOMM, "host exchange callback failed");
        if (d->n_halo) HIPCHK(h, hipMemcpyAsync(x + n_loc, d->host_recv, (size_t)d->n_halo * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (split) { if (nb_bnd) launch_spmv_exact_desc(A, desc_bnd, nb_bnd, a, epi, 1, h->stream); }
        else launch_spmv_reordered(h, A, a, epi);
    }
    HIPCHK(h, hipGetLastError());
    return ITS_OK;
}

// in-place all-reduce of `count` device doubles on the compute stream; op 0 = sum, 1 = max
int its_dist_allreduce(its_handle *h, double *dev, int count, int op)
{
    DistPlan *d = h->dist;
    if (d->transport == 1) {
        NCCLCHK(h, ncclAllReduce(dev, dev, count, ncclDouble, op ? ncclMax : ncclSum, d->comm_red, h->stream));
        return ITS_OK;
    }
    if (count > 2 * ITS_NPART) FAIL(h, ITS_EINVAL, "host all-reduce limited to 2 ITS_NPART values");
    HIPCHK(h, hipMemcpyAsync(d->host_sc, dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (d->rcb(d->user, d->host_sc, count, op) != 0) FAIL(h, ITS_ECOMM, "host all-reduce callback failed");
    HIPCHK(h, hipMemcpyAsync(dev, d->host_sc, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ITS_OK;
}
