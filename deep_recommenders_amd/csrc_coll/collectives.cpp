// include/dr_collectives.h over RCCL (xGMI inside a node).  Host C++, no device code: every call enqueues RCCL work on the
// caller's stream.  The variable all-to-all is a grouped set of point-to-point sends / receives -- on MI355X's full xGMI mesh
// each (sender, receiver) pair owns a link, so the per-peer messages of one exchange travel in parallel.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>
#include <new>
#include "../../include/dr_collectives.h"

namespace {
struct Comm {
    ncclComm_t nccl;
    int32_t world, rank;
};
thread_local char g_err[256] = "";
int fail(ncclResult_t r, const char* what) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, ncclGetErrorString(r));
    return DRC_ERCCL;
}
#define RCCL_TRY(call, what)                       \
    do {                                           \
        ncclResult_t r__ = (call);                 \
        if (r__ != ncclSuccess) return fail(r__, what); \
    } while (0)
static_assert(sizeof(ncclUniqueId) <= DR_COLL_ID_BYTES, "rendezvous id does not fit DR_COLL_ID_BYTES");
}  // namespace

extern "C" const char* dr_coll_last_error(void) { return g_err; }

extern "C" int dr_coll_unique_id(void* id_bytes) {
    if (!id_bytes) return DRC_EINVAL;
    ncclUniqueId id;
    RCCL_TRY(ncclGetUniqueId(&id), "ncclGetUniqueId");
    std::memset(id_bytes, 0, DR_COLL_ID_BYTES);
    std::memcpy(id_bytes, &id, sizeof(id));
    return DRC_OK;
}

extern "C" int dr_coll_init(dr_comm_t* comm, int32_t world, int32_t rank, const void* id_bytes) {
    if (!comm || !id_bytes || world < 1 || rank < 0 || rank >= world) return DRC_EINVAL;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    Comm* c = new (std::nothrow) Comm{nullptr, world, rank};
    if (!c) return DRC_EINVAL;
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(r, "ncclCommInitRank");
    }
    *comm = c;
    return DRC_OK;
}

extern "C" int dr_coll_destroy(dr_comm_t comm) {
    if (!comm) return DRC_EINVAL;
    Comm* c = static_cast<Comm*>(comm);
    ncclResult_t r = ncclCommDestroy(c->nccl);
    delete c;
    return r == ncclSuccess ? DRC_OK : fail(r, "ncclCommDestroy");
}

extern "C" int32_t dr_coll_world(dr_comm_t comm) { return comm ? static_cast<Comm*>(comm)->world : 0; }
extern "C" int32_t dr_coll_rank(dr_comm_t comm) { return comm ? static_cast<Comm*>(comm)->rank : -1; }

// Inside an open ncclGroupStart a failing call must still be followed by ncclGroupEnd, or the communicator stays in an open group
// and every later collective queues behind it: errors are remembered, the group is always closed.
#define RCCL_IN_GROUP(call, what)                                  \
    do {                                                           \
        if (rc == DRC_OK) {                                        \
            ncclResult_t r__ = (call);                             \
            if (r__ != ncclSuccess) rc = fail(r__, what);          \
        }                                                          \
    } while (0)

extern "C" int dr_coll_alltoall_i64(dr_comm_t comm, const int64_t* send, int64_t* recv, int64_t per_peer, drc_stream_t stream) {
    if (!comm || !send || !recv || per_peer < 0) return DRC_EINVAL;
    Comm* c = static_cast<Comm*>(comm);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = DRC_OK;
    RCCL_TRY(ncclGroupStart(), "ncclGroupStart");
    for (int p = 0; p < c->world; ++p) {
        RCCL_IN_GROUP(ncclSend(send + (int64_t)p * per_peer, (size_t)per_peer, ncclInt64, p, c->nccl, s), "ncclSend");
        RCCL_IN_GROUP(ncclRecv(recv + (int64_t)p * per_peer, (size_t)per_peer, ncclInt64, p, c->nccl, s), "ncclRecv");
    }
    const ncclResult_t re = ncclGroupEnd();
    if (rc == DRC_OK && re != ncclSuccess) rc = fail(re, "ncclGroupEnd");
    return rc;
}

extern "C" int dr_coll_alltoallv(dr_comm_t comm, const void* send, const int64_t* send_counts, void* recv,
                                 const int64_t* recv_counts, int64_t elem_bytes, drc_stream_t stream) {
    if (!comm || !send_counts || !recv_counts || elem_bytes <= 0) return DRC_EINVAL;
    Comm* c = static_cast<Comm*>(comm);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const char* sp = static_cast<const char*>(send);
    char* rp = static_cast<char*>(recv);
    // every argument is validated BEFORE the group opens
    for (int p = 0; p < c->world; ++p) {
        if (send_counts[p] < 0 || recv_counts[p] < 0) return DRC_EINVAL;
        if ((send_counts[p] > 0 && !sp) || (recv_counts[p] > 0 && !rp)) return DRC_EINVAL;
    }
    int rc = DRC_OK;
    RCCL_TRY(ncclGroupStart(), "ncclGroupStart");
    int64_t so = 0, ro = 0;
    for (int p = 0; p < c->world; ++p) {
        const int64_t sb = send_counts[p] * elem_bytes, rb = recv_counts[p] * elem_bytes;
        if (sb > 0) RCCL_IN_GROUP(ncclSend(sp + so, (size_t)sb, ncclInt8, p, c->nccl, s), "ncclSend");
        if (rb > 0) RCCL_IN_GROUP(ncclRecv(rp + ro, (size_t)rb, ncclInt8, p, c->nccl, s), "ncclRecv");
        so += sb;
        ro += rb;
    }
    const ncclResult_t re = ncclGroupEnd();
    if (rc == DRC_OK && re != ncclSuccess) rc = fail(re, "ncclGroupEnd");
    return rc;
}

extern "C" int dr_coll_allreduce_f32(dr_comm_t comm, float* buf, int64_t n, drc_stream_t stream) {
    if (!comm || n < 0 || (n > 0 && !buf)) return DRC_EINVAL;
    if (n == 0) return DRC_OK;
    Comm* c = static_cast<Comm*>(comm);
    RCCL_TRY(ncclAllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->nccl, static_cast<hipStream_t>(stream)), "ncclAllReduce");
    return DRC_OK;
}

extern "C" int dr_coll_allgather(dr_comm_t comm, const void* send, void* recv, int64_t n_bytes, drc_stream_t stream) {
    if (!comm || n_bytes < 0 || (n_bytes > 0 && (!send || !recv))) return DRC_EINVAL;
    if (n_bytes == 0) return DRC_OK;
    Comm* c = static_cast<Comm*>(comm);
    RCCL_TRY(ncclAllGather(send, recv, (size_t)n_bytes, ncclInt8, c->nccl, static_cast<hipStream_t>(stream)), "ncclAllGather");
    return DRC_OK;
}
