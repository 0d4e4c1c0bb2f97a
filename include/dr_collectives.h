/* dr_collectives.h -- the exchange steps of the row-sharded hot path (C1 id all-to-all, C2 / C3 row and
 * gradient all-to-all, C4 dense-gradient all-reduce, C5 candidate all-gather) behind a plain C ABI over RCCL, so that a host that
 * is not PyTorch can drive the same plan `deep_recommenders_amd/sharded.py` drives through torch.distributed.
 *
 * The reference has no multi-device code at all: nothing here replaces a reference interface; the entry
 * points are the ones section 8b proposed (`dr_shard_plan` is dr_shard_bucket_ids of dr_hotpath.h, which also produces the split
 * sizes these calls take).  One process per GPU; every call is asynchronous on the caller's hipStream_t and returns an int
 * status (0 ok).  Device pointers, element counts, no torch types.
 *
 *   requesting rank                                   owning rank
 *   dr_shard_bucket_ids  -> counts, send_rows, pos
 *   dr_coll_alltoall_i64(counts)                      (split sizes; copy to the host to size the next calls)
 *   dr_coll_alltoallv(send_rows, 8-byte elements)  -> recv_rows
 *                                                     dr_rows_gather(recv_rows) -> rows_buf
 *   got_rows <- dr_coll_alltoallv(rows_buf, 4 * D-byte elements, splits swapped)
 *   dr_emb_pool_fwd(pos as ids, got_rows as table) ... tower ... dr_emb_pack_grads -> g_rows
 *   dr_coll_alltoallv(g_rows)                      -> g_pad ; dr_emb_pool_bwd_sorted on the owner
 *   dr_coll_allreduce_f32(flat dense gradients)
 */
#ifndef DR_COLLECTIVES_H_
#define DR_COLLECTIVES_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRC_OK 0
#define DRC_EINVAL (-1)   /* bad argument */
#define DRC_ERCCL (-20)   /* an RCCL call failed (dr_coll_last_error() has its text) */

typedef void* dr_comm_t;     /* opaque: an RCCL communicator + its world size / rank */
typedef void* drc_stream_t;  /* hipStream_t */

#define DR_COLL_ID_BYTES 128
/* Rank 0 creates the rendezvous id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by any side channel (MPI,
 * a file, a TCP store); every rank then calls dr_coll_init with the same bytes. */
int dr_coll_unique_id(void* id_bytes);
int dr_coll_init(dr_comm_t* comm, int32_t world, int32_t rank, const void* id_bytes);
int dr_coll_destroy(dr_comm_t comm);
int32_t dr_coll_world(dr_comm_t comm);
int32_t dr_coll_rank(dr_comm_t comm);
const char* dr_coll_last_error(void);

/* Fixed-size all-to-all of int64: rank r receives send[r * per_peer .. (r + 1) * per_peer) of every peer (the split sizes). */
int dr_coll_alltoall_i64(dr_comm_t comm, const int64_t* send, int64_t* recv, int64_t per_peer, drc_stream_t stream);
/* Variable all-to-all of `elem_bytes`-byte elements (8 for row ids, 4 * D for rows / row gradients, 4 for first-order values):
 * send_counts[p] elements go to peer p from consecutive positions of `send`; recv_counts[p] arrive from peer p into consecutive
 * positions of `recv`.  The count arrays are HOST arrays of `world` entries. */
int dr_coll_alltoallv(dr_comm_t comm, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts,
                      int64_t elem_bytes, drc_stream_t stream);
int dr_coll_allreduce_f32(dr_comm_t comm, float* buf, int64_t n, drc_stream_t stream);   /* in place, sum */
/* recv[p * n_bytes ..] = send of peer p (candidate embeddings / ids of the in-batch softmax; queries of the sharded top-K) */
int dr_coll_allgather(dr_comm_t comm, const void* send, void* recv, int64_t n_bytes, drc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DR_COLLECTIVES_H_ */
