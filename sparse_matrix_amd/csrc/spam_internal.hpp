// Internal declarations shared by the HIP translation units of libspam_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <new>

#include "../../include/spam_hip.h"

namespace spam {

constexpr uint32_t EMPTY_KEY = 0xFFFFFFFFu;  // linprobe's sentinel (linprobe/src/set.rs:45)
constexpr uint32_t HASH_SCAL = 107u;         // linprobe/src/lib.rs:13
constexpr uint32_t MIN_TABLE = 16u;          // linprobe/src/lib.rs:14

// ---- error plumbing ---------------------------------------------------------------------------
int set_error(int code, const char* fmt, ...);
void clear_error();

#define SPAM_HIP_TRY(expr)                                                                            \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) {                                                                           \
      int _c = (_e == hipErrorOutOfMemory) ? SPAM_E_OOM : SPAM_E_HIP;                                 \
      return ::spam::set_error(_c, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                               __LINE__);                                                             \
    }                                                                                                 \
  } while (0)

#define SPAM_TRY(expr)          \
  do {                          \
    int _s = (expr);            \
    if (_s != SPAM_OK) return _s; \
  } while (0)

#define SPAM_CHECK_LAUNCH() SPAM_HIP_TRY(hipGetLastError())

hipStream_t stream();

// Diagnostics (SPAM_SPMV_PROF=4): wall-clock stamps of the analysis phases, each after a stream synchronise.
struct PhaseClock {
  bool on;
  hipStream_t s;
  double t0;
  static double now();
  PhaseClock(bool on_, hipStream_t s_) : on(on_), s(s_), t0(on_ ? now() : 0.0) {}
  void stamp(const char* what) {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    const double t = now();
    fprintf(stderr, "[spam] prepare: %-44s %8.2f ms\n", what, (t - t0) * 1e3);
    t0 = t;
  }
};

// Caching device allocator: freed blocks are kept and handed back to later requests of a similar size, so a
// product repeated on same-sized operands pays hipMalloc (seconds for a 100 GB C) only once — the same
// steady state the reference's Vec::with_capacity gets from the process allocator (mul_hash.rs:119).
// spam_hip_trim() returns everything to the driver.
int dev_alloc(void** p, size_t bytes);
void dev_free(void* p);
void dev_trim();

// RAII device buffer (frees on scope exit unless released)
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { reset(); }
  int alloc(size_t n) {
    reset();
    bytes = n;
    if (n == 0) n = 16;
    return dev_alloc(&p, n);
  }
  void reset() {
    if (p) dev_free(p);
    p = nullptr;
    bytes = 0;
  }
  void* release() {
    void* q = p;
    p = nullptr;
    bytes = 0;
    return q;
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

struct SpmvPlan;

}  // namespace spam

// The opaque handle of the C ABI.
struct spam_csr {
  uint64_t rows = 0, cols = 0, nnz = 0;
  int dtype = SPAM_F64;
  int is_sorted = 0;
  uint64_t* row_ptr = nullptr;    // u64[rows+1]
  uint32_t* row_ptr32 = nullptr;  // u32[rows+1], present iff nnz < 2^32 (SpMV reads 4-byte offsets)
  uint32_t* col_idx = nullptr;    // u32[nnz]
  void* vals = nullptr;           // T[nnz]
  bool owns_row_ptr = false, owns_row_ptr32 = false, owns_col_val = false;
  spam::SpmvPlan* spmv_plan = nullptr;  // built by spam_spmv_prepare or on the first SpMV; read-only afterwards
  std::mutex plan_mu;                   // serialises its construction (handles may be used from several host threads)
  std::mutex run_mu;  // serialises the ENQUEUE of one SpMV's kernels: they share the plan's scratch (tables, queue heads, pair
                      // sums, carries), which is safe in stream order but not when two threads' launches interleave
  // Set by spam_csr_allgather_rccl (spam_rccl.hip): `vals` is being filled by the exchange stream and is complete once this
  // event has fired.  Everything that reads vals on the library stream waits for it first (spam::wait_values); the
  // symbolic phase and the partition read only row_ptr / col_idx and do not, which is how the transfer of the values
  // (2/3 of B's bytes) runs under them.  Row-block views share their parent's event.
  hipEvent_t vals_ready = nullptr;
  bool owns_vals_event = false;
};

namespace spam {

inline size_t dtype_size(int dtype) { return dtype == SPAM_F32 ? 4 : 8; }
// make stream s wait until m's values have arrived (no-op for every handle that was not gathered asynchronously)
inline int wait_values(const spam_csr* m, hipStream_t s) {
  if (m && m->vals_ready) SPAM_HIP_TRY(hipStreamWaitEvent(s, m->vals_ready, 0));
  return SPAM_OK;
}
// SPAM_I64 values are computed on their bit pattern as unsigned 64-bit integers: unsigned arithmetic wraps, which IS
// Wrapping<i64>'s two's-complement behaviour (and keeps the kernels free of signed-overflow undefined behaviour)
typedef unsigned long long i64w;
// What an accumulator holds before the first product arrives.  The reference STORES the first product of a column
// (`or_insert(t1)`, mul_hash.rs:157-161; map.rs:113-116) and adds the later ones; an accumulator that starts at the additive
// identity and adds every product gives the same value — provided the identity is -0.0, not +0.0: (+0.0) + (-0.0) = +0.0
// would flip the sign of a first product of -0.0, (-0.0) + x = x for every x.  Integers: 0.
template <class T>
__host__ __device__ constexpr T acc_identity() { return T(0); }
template <>
__host__ __device__ constexpr double acc_identity<double>() { return -0.0; }
template <>
__host__ __device__ constexpr float acc_identity<float>() { return -0.0f; }
// run f.template operator()<T>() for the value type of a handle
#define SPAM_DISPATCH_DTYPE(dtype, CALL_F64, CALL_F32, CALL_I64) \
  ((dtype) == SPAM_F64 ? (CALL_F64) : (dtype) == SPAM_F32 ? (CALL_F32) : (CALL_I64))

// ---- device primitives (spam_scan.hip) ----------------------------------------------------------
// out[0] = 0, out[i+1] = sum(in[0..i]) for i < n, u64 accumulation (checked_inclusive_scan's shape,
// spam_csr/src/lib.rs:267-274).  tmp must hold scan_tmp_bytes(n).
size_t scan_tmp_bytes(uint64_t n);
int scan_u32_to_u64(const uint32_t* d_in, uint64_t n, uint64_t* d_out, void* d_tmp, hipStream_t s);
int scan_u64_to_u64(const uint64_t* d_in, uint64_t n, uint64_t* d_out, void* d_tmp, hipStream_t s);
// *d_flag |= 1 if any partial sum wrapped (checked_add overflow)
int narrow_row_ptr(const uint64_t* d_in, uint64_t n, uint32_t* d_out, hipStream_t s);

// d_out = d_in sorted in descending order (rocPRIM radix sort; spam_spmv_rg.hip)
int sort_u32_desc(const uint32_t* d_in, uint32_t* d_out, uint64_t n, hipStream_t s);
// d_out = d_in sorted in ascending order on the low `bits` bits (the keys must not use higher ones)
int sort_u32_asc(const uint32_t* d_in, uint32_t* d_out, uint64_t n, int bits, hipStream_t s);

// ---- SpMV (spam_spmv.hip) ------------------------------------------------------------------------
void spmv_plan_free(SpmvPlan* p);
int spmv_launch(const spam_csr* a, const void* d_x, void* d_y, hipStream_t s);

// ---- SpMV, XCD-sliced path: constants shared by its two translation units -----------------------------------------
constexpr int SPMV_NSLAB = 8;               // one table slab per XCD of MI355X
constexpr int SPMV_NPART = SPMV_NSLAB + 1;  // + the cold part: entries whose column is in no table (gathered from x)
constexpr int SPMV_HEAD_STRIDE = 64;        // u32 words between two queue heads: one 256-byte block each, so the
                                            // eight XCDs' dequeue atomics do not serialise on one line

// ---- SpMV, row-group form of the sliced path's long rows (spam_spmv_rg.hip) ------------------------------------------
struct RgKnobs {
  int eblk = 0;       // SPAM_SPMV_RG_EBLK   entries of a (part, row group) block aimed at (0: entries / 500 within 16 k .. 200 k)
  int nt = 512;       // SPAM_SPMV_RG_NT     threads per workgroup (256 / 512 / 1024)
  int k = 4;          // SPAM_SPMV_RG_K      chunks a wavefront keeps in flight
  int acc_bits = 0;   // SPAM_SPMV_RG_ABITS  cap on the accumulator bits of an entry word (0: as many as fit, at most 13)
  int fold_short = 1; // SPAM_SPMV_RG_SHORT  the short rows as blocks of the same kernel (0: spmv_short_rows_kernel)
  int prof = 0;       // SPAM_SPMV_PROF = 3   per-block phase timers, printed by every SpMV (diagnostics)
  int order_min = 32; // SPAM_SPMV_RG_ORDER  rows with fewer entries are ranked in row order (coalesced y writes), longer ones by length
  int keep_vsrc = 0;  // SPAM_SPMV_RG_VSRC   1: build the value-source map (4 bytes per padded entry) at prepare; 0: on the first
                      //                     spam_csr_values_changed (most matrices never change their values)
  int fmt = 2;        // SPAM_SPMV_RG_FMT    packed entry words a block may use: 0 none (32-bit words), 1 up to 24-bit, 2 up to 16-bit
};
struct RgPlan {
  uint32_t ng = 0, abits = 0, lds_bytes = 0, nl = 0, ntiles = 0;
  uint32_t ngs = 0, ns = 0;         // short-row blocks (0: no short rows, or they keep their own kernel)
  int nt = 512, k = 4, prof = 0;
  unsigned grid = 0;
  uint64_t stream_bytes = 0, entries = 0, padded = 0, naccs = 0, npairs = 0, nonempty_blocks = 0;
  unsigned char* stream = nullptr;  // the blocks' chunks, part-major
  void* blk = nullptr;              // RgBlk[NPART * ng + ngs + 1]: block descriptors (plan layout: fill / refresh)
  void* qblk = nullptr;             // RgBlk[nq]: the kernel's queue entries (a block, or a piece of one), queue by queue
  uint32_t nq = 0, qbase[SPMV_NPART + 1] = {0}, qcount[SPMV_NPART + 1] = {0};
  uint64_t* pe = nullptr;           // u64[blocks + 1]: padded entries before every block
  uint32_t* sub_head = nullptr;     // u32[naccs - npairs]: head of every extra accumulator
  uint32_t* slot_rows = nullptr;    // u32[64 ntiles]: row of every rank slot (0xFFFFFFFF: none)
  uint16_t* mask_slot = nullptr;    // u16[64 ntiles]: parts in which the row of a slot has a pair
  void* tiles = nullptr;            // RgTile[ntiles]: 64 rank slots of one group each
  uint64_t* grp_cp = nullptr;       // u64[ng][NPART]: first pair of every (group, part) block in cp
  void* cp = nullptr;               // T[npairs]: the pair sums of one SpMV
  uint32_t* vsrc = nullptr;         // u32[padded]: source entry of every value position (spam_csr_values_changed); null until
                                    // the first such call unless SPAM_SPMV_RG_VSRC=1 (0.69 GB for the bench matrix)
  uint64_t* piece_pairs = nullptr;  // u64[n_piece_pairs]: pairs of the blocks run in pieces (their sums meet by atomic adds):
  uint32_t n_piece_pairs = 0;       // set to zero by the first kernel of every product
};
void rg_plan_free(RgPlan* g);
// *out stays null when the matrix does not qualify (the caller then builds the tile form)
template <class T>
int rg_build(const spam_csr* a, const uint8_t* slab, const uint32_t* code, const uint32_t* long_rows, uint32_t nl, uint32_t S,
             const spam_csr* res, uint32_t ns, const RgKnobs& kn, hipStream_t s, RgPlan** out);
template <class T>
int rg_run(const RgPlan* g, const T* xh, uint32_t S, const T* x, uint32_t* queue, T* y, const uint32_t* short_rows,
           hipStream_t s);
template <class T>
int rg_refresh_values(const spam_csr* a, const spam_csr* res, RgPlan* g, hipStream_t s);

}  // namespace spam
