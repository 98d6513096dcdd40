// spam_rccl.hip — libspam_hip_rccl.so: the multi-GPU exchange behind the C ABI (include/spam_hip_rccl.h).
//
// Reference counterpart: none as code — the reference's threads share B through memory (mul_hash.rs:72-76, 130).  What
// is carried over is the partition rule (rows_to_threads, mul_hash.rs:38-64, tnum = ranks) and the ownership of C (each
// task writes a disjoint slice, mul_hash.rs:119-128).  One process per GPU; the all-gather-v of B is a grouped
// ncclSend / ncclRecv per peer, so each of the 7 point-to-point xGMI links of an MI355X carries one shard.
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/spam_hip_rccl.h"
#include "spam_internal.hpp"

struct spam_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  bool owns_comm = false;
  hipStream_t xs = nullptr;      // exchange stream
  hipEvent_t ev_in = nullptr;    // library stream -> exchange stream
  hipEvent_t ev_out = nullptr;   // exchange stream -> library stream (structure / vectors)
  uint64_t last_recv = 0, last_sent = 0;
};

namespace spam {

#define SPAM_NCCL_TRY(expr)                                                                                        \
  do {                                                                                                             \
    ncclResult_t _r = (expr);                                                                                      \
    if (_r != ncclSuccess)                                                                                         \
      return ::spam::set_error(SPAM_E_COMM, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

// out[i] = in[i] - base + add for i < n; tail[0] = tail_value when tail is given (the gathered row_ptr's last entry)
__global__ void rebase_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n, uint64_t base,
                              uint64_t add, uint64_t* __restrict__ tail, uint64_t tail_value) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  if (i == 0 && tail) *tail = tail_value;
  for (; i < n; i += stride) out[i] = in[i] - base + add;
}

static int comm_finish_init(spam_comm* c) {
  SPAM_HIP_TRY(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
  SPAM_HIP_TRY(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
  SPAM_HIP_TRY(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
  return SPAM_OK;
}

// exchange stream waits for everything queued on the library stream so far
static int order_after_library(spam_comm* c) {
  SPAM_HIP_TRY(hipEventRecord(c->ev_in, stream()));
  SPAM_HIP_TRY(hipStreamWaitEvent(c->xs, c->ev_in, 0));
  return SPAM_OK;
}
// library stream waits for everything queued on the exchange stream so far
static int library_after_exchange(spam_comm* c) {
  SPAM_HIP_TRY(hipEventRecord(c->ev_out, c->xs));
  SPAM_HIP_TRY(hipStreamWaitEvent(stream(), c->ev_out, 0));
  return SPAM_OK;
}

// One all-gather-v on the exchange stream: this rank's `mine` (bytes[rank] bytes) goes to every rank, rank r's piece
// arrives at full + start[r].  Every pair is one ncclSend + one ncclRecv inside one group: all peers at once, one shard
// per link.  Empty pieces are skipped on both sides (every rank knows every size).
static int allgatherv_bytes(spam_comm* c, const void* mine, void* full, const std::vector<uint64_t>& start /* world+1 */) {
  const int W = c->world, me = c->rank;
  const uint64_t my_bytes = start[me + 1] - start[me];
  SPAM_NCCL_TRY(ncclGroupStart());
  for (int r = 0; r < W; ++r) {
    const uint64_t nb = start[r + 1] - start[r];
    if (my_bytes) SPAM_NCCL_TRY(ncclSend(mine, my_bytes, ncclUint8, r, c->comm, c->xs));
    if (nb) SPAM_NCCL_TRY(ncclRecv((char*)full + start[r], nb, ncclUint8, r, c->comm, c->xs));
    if (r != me) c->last_recv += nb, c->last_sent += my_bytes;
  }
  SPAM_NCCL_TRY(ncclGroupEnd());
  return SPAM_OK;
}

static int check_comm(const spam_comm* c) {
  if (!c || !c->comm) return set_error(SPAM_E_INVALID, "communicator is NULL");
  return SPAM_OK;
}

}  // namespace spam

using namespace spam;

extern "C" {

int spam_rccl_unique_id(void* id_out) {
  clear_error();
  if (!id_out) return set_error(SPAM_E_INVALID, "id_out is NULL");
  static_assert(sizeof(ncclUniqueId) == SPAM_RCCL_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  SPAM_NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return SPAM_OK;
}

int spam_rccl_comm_create(const void* id_bytes, int rank, int world, spam_comm** out) {
  clear_error();
  if (!id_bytes || !out) return set_error(SPAM_E_INVALID, "NULL argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return set_error(SPAM_E_INVALID, "rank %d outside world %d", rank, world);
  spam_comm* c = new (std::nothrow) spam_comm();
  if (!c) return set_error(SPAM_E_OOM, "host allocation failed");
  c->rank = rank, c->world = world, c->owns_comm = true;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof id);
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return set_error(SPAM_E_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
  }
  if (int e = comm_finish_init(c)) {
    spam_rccl_comm_free(c);
    return e;
  }
  *out = c;
  return SPAM_OK;
}

int spam_rccl_comm_adopt(void* nccl_comm, int rank, int world, spam_comm** out) {
  clear_error();
  if (!nccl_comm || !out) return set_error(SPAM_E_INVALID, "NULL argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return set_error(SPAM_E_INVALID, "rank %d outside world %d", rank, world);
  int n = 0, u = 0;
  SPAM_NCCL_TRY(ncclCommCount((ncclComm_t)nccl_comm, &n));
  SPAM_NCCL_TRY(ncclCommUserRank((ncclComm_t)nccl_comm, &u));
  if (n != world || u != rank) return set_error(SPAM_E_INVALID, "communicator is rank %d of %d, not %d of %d", u, n, rank, world);
  spam_comm* c = new (std::nothrow) spam_comm();
  if (!c) return set_error(SPAM_E_OOM, "host allocation failed");
  c->comm = (ncclComm_t)nccl_comm, c->rank = rank, c->world = world, c->owns_comm = false;
  if (int e = comm_finish_init(c)) {
    spam_rccl_comm_free(c);
    return e;
  }
  *out = c;
  return SPAM_OK;
}

int spam_rccl_comm_info(const spam_comm* c, int* rank, int* world) {
  clear_error();
  SPAM_TRY(check_comm(c));
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return SPAM_OK;
}

int spam_rccl_last_exchange(const spam_comm* c, uint64_t* bytes_received, uint64_t* bytes_sent) {
  clear_error();
  SPAM_TRY(check_comm(c));
  if (bytes_received) *bytes_received = c->last_recv;
  if (bytes_sent) *bytes_sent = c->last_sent;
  return SPAM_OK;
}

void spam_rccl_comm_free(spam_comm* c) {
  if (!c) return;
  if (c->xs) (void)hipStreamSynchronize(c->xs);
  if (c->comm && c->owns_comm) (void)ncclCommDestroy(c->comm);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->xs) (void)hipStreamDestroy(c->xs);
  delete c;
}

int spam_csr_allgather_rccl(spam_comm* c, const spam_csr* shard, const uint64_t* row_offsets, spam_csr** full_out) {
  clear_error();
  SPAM_TRY(check_comm(c));
  if (!shard || !row_offsets || !full_out) return set_error(SPAM_E_INVALID, "NULL argument");
  *full_out = nullptr;
  const int W = c->world, me = c->rank;
  if (row_offsets[0] != 0) return set_error(SPAM_E_INVALID, "row_offsets[0] != 0");
  for (int r = 0; r < W; ++r)
    if (row_offsets[r + 1] < row_offsets[r]) return set_error(SPAM_E_INVALID, "row_offsets not ascending at rank %d", r);
  const uint64_t rows = row_offsets[W], my_rows = row_offsets[me + 1] - row_offsets[me];
  if (rows == 0 || rows >= 0xFFFFFFFFull) return set_error(SPAM_E_INVALID, "gathered matrix would have %llu rows", (unsigned long long)rows);
  if (shard->rows != my_rows)
    return set_error(SPAM_E_INVALID, "shard has %llu rows, row_offsets give rank %d %llu", (unsigned long long)shard->rows, me,
                     (unsigned long long)my_rows);
  hipStream_t s = stream();
  const size_t vs = dtype_size(shard->dtype);
  c->last_recv = c->last_sent = 0;

  // the shard's entries are [rp[0], rp[rows]) of its arrays (a row-block view starts up to 3 entries early)
  uint64_t ends[2];
  SPAM_HIP_TRY(hipMemcpyAsync(&ends[0], shard->row_ptr, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(&ends[1], shard->row_ptr + shard->rows, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  const uint64_t rp0 = ends[0], my_nnz = ends[1] - ends[0];

  // shard sizes: 8 bytes per rank
  DevBuf d_sizes;
  SPAM_TRY(d_sizes.alloc((size_t)(W + 1) * 8));
  uint64_t* dsz = d_sizes.as<uint64_t>();
  SPAM_TRY(order_after_library(c));
  SPAM_HIP_TRY(hipMemcpyAsync(dsz + W, &my_nnz, 8, hipMemcpyHostToDevice, c->xs));
  SPAM_NCCL_TRY(ncclAllGather(dsz + W, dsz, 1, ncclUint64, c->comm, c->xs));
  std::vector<uint64_t> nnz_of(W);
  SPAM_HIP_TRY(hipMemcpyAsync(nnz_of.data(), dsz, (size_t)W * 8, hipMemcpyDeviceToHost, c->xs));
  SPAM_HIP_TRY(hipStreamSynchronize(c->xs));
  std::vector<uint64_t> start(W + 1, 0);
  for (int r = 0; r < W; ++r) start[r + 1] = start[r] + nnz_of[r];
  const uint64_t nnz = start[W];
  if (nnz_of[me] != my_nnz) return set_error(SPAM_E_COMM, "size exchange returned %llu for this rank's %llu entries", (unsigned long long)nnz_of[me], (unsigned long long)my_nnz);

  spam_csr* m = new (std::nothrow) spam_csr();
  if (!m) return set_error(SPAM_E_OOM, "host allocation failed");
  m->rows = rows, m->cols = shard->cols, m->nnz = nnz, m->dtype = shard->dtype, m->is_sorted = shard->is_sorted;
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(c->xs);
    spam_csr_free(m);
    return code;
  };
  DevBuf rp, ci, va, my_rp;
  if (int e = rp.alloc((rows + 1) * 8)) return fail(e);
  if (int e = ci.alloc(nnz * 4)) return fail(e);
  if (int e = va.alloc(nnz * vs)) return fail(e);
  if (int e = my_rp.alloc(my_rows * 8)) return fail(e);
  m->row_ptr = (uint64_t*)rp.release();
  m->col_idx = (uint32_t*)ci.release();
  m->vals = va.release();
  m->owns_row_ptr = m->owns_col_val = true;
#define AG_TRY(expr)          \
  do {                        \
    int _e = (expr);          \
    if (_e) return fail(_e);  \
  } while (0)
#define AG_HIP(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t _h = (expr);                                                                               \
    if (_h != hipSuccess) return fail(set_error(SPAM_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_h))); \
  } while (0)
  // row_ptr fix-up on the sender: local offsets -> offsets into the gathered arrays; every rank writes the final entry
  {
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((my_rows + 255) / 256, 4096));
    rebase_kernel<<<grid, 256, 0, s>>>(shard->row_ptr, my_rp.as<uint64_t>(), my_rows, rp0, start[me], m->row_ptr + rows, nnz);
    AG_HIP(hipGetLastError());
  }
  AG_TRY(order_after_library(c));
  if (shard->vals_ready) AG_HIP(hipStreamWaitEvent(c->xs, shard->vals_ready, 0));
  std::vector<uint64_t> b_rp(W + 1), b_ci(W + 1), b_va(W + 1);
  for (int r = 0; r <= W; ++r) b_rp[r] = row_offsets[r] * 8, b_ci[r] = start[r] * 4, b_va[r] = start[r] * vs;
  AG_TRY(allgatherv_bytes(c, my_rp.p, m->row_ptr, b_rp));
  AG_TRY(allgatherv_bytes(c, shard->col_idx + rp0, m->col_idx, b_ci));
  AG_TRY(library_after_exchange(c));  // structure complete: the library stream may read row_ptr / col_idx
  AG_TRY(allgatherv_bytes(c, (const char*)shard->vals + rp0 * vs, m->vals, b_va));
  AG_HIP(hipEventCreateWithFlags(&m->vals_ready, hipEventDisableTiming));
  m->owns_vals_event = true;
  AG_HIP(hipEventRecord(m->vals_ready, c->xs));
  // my_rp goes back to the allocator here: every later use of that block is queued on the library stream behind ev_out
  if (nnz < (1ull << 32)) {
    DevBuf r32;
    AG_TRY(r32.alloc((rows + 1) * 4));
    m->row_ptr32 = (uint32_t*)r32.release();
    m->owns_row_ptr32 = true;
    AG_TRY(narrow_row_ptr(m->row_ptr, rows + 1, m->row_ptr32, s));
  }
#undef AG_TRY
#undef AG_HIP
  *full_out = m;
  return SPAM_OK;
}

int spam_vector_allgather_rccl(spam_comm* c, int elem_bytes, const void* d_local, const uint64_t* counts, void* d_full) {
  clear_error();
  SPAM_TRY(check_comm(c));
  if (!counts || !d_full) return set_error(SPAM_E_INVALID, "NULL argument");
  if (elem_bytes != 4 && elem_bytes != 8) return set_error(SPAM_E_INVALID, "elem_bytes must be 4 or 8");
  const int W = c->world;
  if (counts[c->rank] && !d_local) return set_error(SPAM_E_INVALID, "d_local is NULL");
  std::vector<uint64_t> start(W + 1, 0);
  for (int r = 0; r < W; ++r) start[r + 1] = start[r] + counts[r] * (uint64_t)elem_bytes;
  c->last_recv = c->last_sent = 0;
  SPAM_TRY(order_after_library(c));
  SPAM_TRY(allgatherv_bytes(c, d_local, d_full, start));
  SPAM_TRY(library_after_exchange(c));
  return SPAM_OK;
}

int spam_spgemm_rank_step_rccl(spam_comm* c, const spam_csr* a_full, const spam_csr* b_full, const uint64_t* count_offsets,
                               uint64_t row_overhead, int sort_rows, spam_csr** c_block, uint64_t* row_lo, uint64_t* row_hi,
                               spam_spgemm_stats* stats) {
  clear_error();
  SPAM_TRY(check_comm(c));
  if (!a_full || !b_full || !c_block || !row_lo || !row_hi) return set_error(SPAM_E_INVALID, "NULL argument");
  *c_block = nullptr;
  const int W = c->world, me = c->rank;
  const uint64_t rows = a_full->rows;
  std::vector<uint64_t> co(W + 1);
  for (int r = 0; r <= W; ++r) co[r] = count_offsets ? count_offsets[r] : rows * (uint64_t)r / (uint64_t)W;
  if (co[0] != 0 || co[W] != rows) return set_error(SPAM_E_INVALID, "count_offsets must run from 0 to rows(A)");
  for (int r = 0; r < W; ++r)
    if (co[r + 1] < co[r]) return set_error(SPAM_E_INVALID, "count_offsets not ascending at rank %d", r);
  // (1) this rank's share of the flop count (mul_hash.rs:39-50): reads A's row range and B's row_ptr only
  DevBuf mine, flops;
  SPAM_TRY(mine.alloc((co[me + 1] - co[me]) * 8));
  SPAM_TRY(flops.alloc(rows * 8));
  SPAM_TRY(spam_row_flops_dev(a_full, b_full, co[me], co[me + 1], mine.as<uint64_t>()));
  // (2) all-gather the per-row counts: 8 bytes per row of A
  std::vector<uint64_t> start(W + 1);
  for (int r = 0; r <= W; ++r) start[r] = co[r] * 8;
  c->last_recv = c->last_sent = 0;
  SPAM_TRY(order_after_library(c));
  SPAM_TRY(allgatherv_bytes(c, mine.p, flops.p, start));
  SPAM_TRY(library_after_exchange(c));
  if (row_overhead == SPAM_ROW_OVERHEAD_AUTO) row_overhead = spam_row_overhead_auto(b_full);
  // (3) the partition rule on the same vector on every rank (mul_hash.rs:51-62, tnum = world)
  std::vector<uint64_t> offs(W + 1);
  SPAM_TRY(spam_rows_to_parts_from_flops_dev(flops.as<uint64_t>(), rows, (uint64_t)W, row_overhead, offs.data()));
  *row_lo = offs[me], *row_hi = offs[me + 1];
  if (stats) *stats = spam_spgemm_stats{};
  if (*row_hi <= *row_lo) return SPAM_OK;
  // (4) the block's two-phase product; the numeric phase is the first reader of gathered values
  spam_csr* blk = nullptr;
  SPAM_TRY(spam_csr_row_block(a_full, *row_lo, *row_hi, &blk));
  const int e = spam_spgemm_dev(blk, b_full, sort_rows, c_block, stats);
  spam_csr_free(blk);
  return e;
}

}  // extern "C"
