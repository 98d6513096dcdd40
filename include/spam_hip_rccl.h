/* spam_hip_rccl.h — C ABI of libspam_hip_rccl.so: the multi-GPU exchange of the spam_csr SpGEMM / SpMV path, one process
 * (one rank) per MI355X, RCCL over xGMI.  Sibling of libspam_hip.so (spam_hip.h): it links librccl and libspam_hip.so,
 * so that hosts which never leave one GPU do not load RCCL.
 *
 * What this stands in for.  The reference is one process with one thread pool; its only partition is rows_to_threads
 * (spam_csr/src/mul_hash.rs:38-64): contiguous row ranges of A balanced by intermediate products, one per thread, every
 * thread reading all of B through shared memory (mul_hash.rs:72-76, 130).  Across GPUs the same rule cuts A's rows
 * (tnum = ranks); "every thread reads all of B" becomes ONE exchange — an all-gather-v of B's three CSR arrays — and
 * nothing is reduced: C stays row-sharded, exactly as each thread of the reference owns a disjoint slice of the output
 * (mul_hash.rs:119-128).  A Rust host binds these entry points from the same `extern "C"` block as spam_hip.h's
 * (INTEGRATION.md §5 has the call sequence); it needs no NCCL binding of its own: the unique id is 128 opaque bytes it
 * moves over whatever channel launched the ranks.
 *
 * Streams.  The library owns one exchange stream per communicator.  Every call orders it after the work already queued
 * on the library stream (spam_hip_set_stream) and makes the library stream wait for exactly what the next kernels read:
 * the STRUCTURE (row_ptr, col_idx) of a gathered matrix at once, its VALUES only in front of the first kernel that reads
 * them (the handle carries the event) — so the partition, the flop count and the whole symbolic phase run while the
 * values, 2/3 of the bytes, are still crossing xGMI.  No call blocks the host on a transfer except where it must read a
 * size back (the shard sizes; the partition offsets).
 *
 * Pattern on the wire.  xGMI is point-to-point (7 links per GPU), so the all-gather-v is a grouped ncclSend / ncclRecv
 * per peer — every link carries exactly one shard, concurrently, no ring, no padding: bytes received = the unpadded
 * size of the other ranks' shards.
 *
 * Errors: as spam_hip.h (status + spam_last_error()); RCCL failures are SPAM_E_COMM with ncclGetErrorString's text.
 * A collective that a peer never joins blocks, as every NCCL collective does. */
#ifndef SPAM_HIP_RCCL_H
#define SPAM_HIP_RCCL_H

#include "spam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPAM_RCCL_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct spam_comm spam_comm; /* one rank's end of a communicator + its exchange stream and events */

/* ncclGetUniqueId: called by ONE rank; the host hands the 128 bytes to every other rank (file, socket, env, MPI ...). */
int spam_rccl_unique_id(void* id_out /* SPAM_RCCL_ID_BYTES */);
/* ncclCommInitRank on the device selected by spam_hip_init; collective over the `world` ranks. */
int spam_rccl_comm_create(const void* id /* SPAM_RCCL_ID_BYTES */, int rank, int world, spam_comm** out);
/* Wrap a communicator the host already has (an ncclComm_t from its own NCCL binding); not destroyed by comm_free. */
int spam_rccl_comm_adopt(void* nccl_comm, int rank, int world, spam_comm** out);
int spam_rccl_comm_info(const spam_comm* c, int* rank, int* world);
/* Bytes this rank received / sent in the last exchange call (all-gather of a matrix, of a vector, of the flop counts). */
int spam_rccl_last_exchange(const spam_comm* c, uint64_t* bytes_received, uint64_t* bytes_sent);
void spam_rccl_comm_free(spam_comm* c);

/* All-gather-v of a row-sharded CsrMatrix: rank r holds rows [row_offsets[r], row_offsets[r+1]) as `shard` (its own
 * matrix: row_ptr local to the shard — an upload, a spam_csr_from_device wrap or a spam_csr_row_block view); every
 * rank gets *full_out = the whole matrix (owned by the handle; free with spam_csr_free), row_ptr rebased by each
 * source's entry offset on the device.  Order on the wire: shard sizes (8 bytes per rank), row_ptr, col_idx, then vals.
 * On return the library stream waits for row_ptr / col_idx; *full_out may be passed to any entry point at once — those
 * that read values (numeric phase, SpMV, download, add ...) wait for them on the device, spam_spgemm_symbolic,
 * spam_rows_to_parts* and spam_row_flops_dev do not.  row_offsets: world + 1 ascending values, the same on every rank,
 * row_offsets[0] = 0.  All ranks must pass shards of one dtype, cols and sortedness. */
int spam_csr_allgather_rccl(spam_comm* c, const spam_csr* shard, const uint64_t* row_offsets /* world+1 */,
                            spam_csr** full_out);

/* All-gather-v of a dense vector (row-sharded SpMV needs all of x; rank r owns counts[r] elements, concatenated in rank
 * order into d_full).  elem_bytes = 4 or 8.  The library stream waits for the result. */
int spam_vector_allgather_rccl(spam_comm* c, int elem_bytes, const void* d_local, const uint64_t* counts /* world */,
                               void* d_full);

/* One rank's share of C = A B on `world` GPUs — spam_spgemm_rank_step (spam_hip.h) with the per-rank fixed work cut to
 * the rank's share: the flop count (mul_hash.rs:39-50) runs only over rows [count_offsets[rank], count_offsets[rank+1])
 * of A (NULL: equal row blocks) and the per-row counts, 8 bytes per row, are all-gathered; then every rank applies the
 * partition rule (mul_hash.rs:51-62, tnum = world, on F_i + row_overhead * [F_i > 0]) to the same vector, takes its
 * block [*row_lo, *row_hi) as a zero-copy view and runs the two-phase product against all of B.  a_full / b_full: this
 * rank's complete copies (for C = A A one handle, straight from spam_csr_allgather_rccl — its values may still be in
 * flight: only the numeric phase waits for them).  *c_block = NULL when the block has no rows. */
int spam_spgemm_rank_step_rccl(spam_comm* c, const spam_csr* a_full, const spam_csr* b_full,
                               const uint64_t* count_offsets /* world+1 or NULL */, uint64_t row_overhead, int sort_rows,
                               spam_csr** c_block, uint64_t* row_lo, uint64_t* row_hi, spam_spgemm_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* SPAM_HIP_RCCL_H */
