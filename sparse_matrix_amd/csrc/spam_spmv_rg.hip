// SpMV, XCD-sliced path, "row-group" form (gfx950 only).
//
// The XCD-sliced analysis (spam_spmv.hip) deals the most referenced columns over eight table slabs, one per XCD's L2,
// and cuts every sliced row into nine PARTS: its entries in slab 0..7 and the cold rest.  The first form of that path
// kept each part as a CSR over the sliced rows and summed rows tile by tile: the x gathers of a tile go to unrelated
// table lines, every lane costs the L2 one request, and 245 G requests/s bound the kernel (0.78 ms for 154 M entries).
//
// This form turns the part around.  The sliced rows are ranked (32 entries and more: by length, descending; the others
// behind them in row order) and cut into ROW GROUPS; a BLOCK is (part, row group):
//   * the block's accumulators live in LDS (8 bytes each: f64 for f32 / f64 values, u64 for i64), one per (row, part)
//     PAIR that exists — a row without entries in the part has none, so short rows cost what they hold;
//   * the block's entries are stored in TABLE-SLOT order, 256-entry chunks, each entry one 32-bit word
//     (slot << abits | accumulator) and its value: a wave instruction gathers 64 consecutive slots — a handful of
//     128-byte lines that the CU's L1 serves — and adds its products with LDS atomics (ds_add_f64 / ds_add_u64);
//   * the block stores its sums as one contiguous run of the pair array `cp` (pairs of a block in rank order);
//   * spmv_rg_reduce_kernel, one wavefront per TILE of 64 rank slots (a group starts on a multiple of 64 slots), finds
//     a row's pair in every part from the row's part mask (position = tile base + ballot prefix), adds them in part
//     order and writes y[row].
// Rows that would draw most lanes of a wave instruction to one LDS address get extra accumulators behind the block's
// heads (an entry picks one by its position in the row); they are folded into their head before the sums are stored.
// Rows below the sliced-row threshold (by default: rows of one or two entries) join as a tenth kind of block: consecutive short rows, all
// their entries unsliced in column order, gathered from x itself, results written straight to y.
//
// The kernel runs at the fabric's byte rate: what a change saves is the bytes it no longer moves.
//
// Sums of f32 / f64 values are accumulated by LDS atomics in an order that varies from run to run: results are
// reproducible to rounding, not bit for bit (i64 results are exact and bit-identical).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>  // device radix sort of the analysis (AMD's native primitives; round 3
#include <rocprim/types/double_buffer.hpp>      // went through hipCUB, the CUB-compatibility layer on top of them)

#include <algorithm>
#include <vector>

#include "spam_internal.hpp"

namespace spam {

namespace {

constexpr uint32_t RG_NONE = 0xFFFFFFFFu;
constexpr int RG_KEY_ACC_BITS = 14, RG_KEY_IDX_BITS = 32;
constexpr int RG_KEY_BLK_SHIFT = RG_KEY_ACC_BITS + RG_KEY_IDX_BITS;  // 46
constexpr uint32_t RG_MAX_SUB = 512;
constexpr uint32_t RG_FULL_MASK = (1u << SPMV_NPART) - 1u;
constexpr uint32_t RG_SHORT = (uint32_t)SPMV_NPART;  // queue / part number of the short-row blocks

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define RG_GLB __attribute__((address_space(1)))

template <class T>
struct RgAcc {
  typedef double type;
  static __device__ __forceinline__ double prod(T v, T x) { return (double)v * (double)x; }
};
template <>
struct RgAcc<i64w> {
  typedef i64w type;
  static __device__ __forceinline__ i64w prod(i64w v, i64w x) { return v * x; }
};

__device__ __forceinline__ uint32_t rg_xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7u;
}

template <class T>
__host__ __device__ constexpr uint32_t rg_chunk_bytes(bool wide) {
  return 1024u + (wide ? 512u : 0u) + 256u * (uint32_t)sizeof(T);
}
// PACKED chunk formats of the slab parts (round 4).  A block's entries are sorted by table slot, so inside a chunk the
// slot of sorted entry e is the slot of entry e - 1 plus a small DELTA (a 200 k-entry block walks a 347 k-slot table:
// 1.7 slots per entry); the entry word then needs the 13 accumulator bits and a few delta bits instead of a 19-bit slot:
//   fmt 0: 32-bit words, slot << abits | accumulator (the plain format above), 12 bytes per f64 entry;
//   fmt 1: 24-bit words, delta (11 bits) << 13 | accumulator, 11 bytes per f64 entry;
//   fmt 2: 16-bit words, delta (3 bits) << 13 | accumulator, 10 bytes per f64 entry (6 per f32 entry instead of 8).
// The absolute slot of a packed chunk's first entry (whose own delta field is 0) is kept out of line, in a header array at
// the start of the block (one u32 per chunk, padded to 128 bytes), so that every chunk is a whole number of 128-byte
// lines (2560 / 2816 bytes for f64) and starts on one — with the header inside, every chunk straddled two lines that its
// neighbours, streamed by other wavefronts with non-temporal loads, fetched again: 6 % slower than the plain format
// instead of faster.  Lane l's four words are bytes [4 b l, 4 b (l + 1)) of the chunk (b = 2 or 3 bytes per word), its
// k-th word sorted entry 64 k + l as before, so a gather instruction still covers 64 consecutive sorted entries.  A delta that does not
// fit is bridged by PADDING entries (spare accumulator, value 0, delta 7 / 2047); rg_build picks per block the format
// with the fewest bytes (sparse blocks keep fmt 0).  The slots are rebuilt by four wave prefix sums per chunk (DPP).
constexpr uint32_t RG_FMT_SHIFT = 30, RG_NCHUNK_MASK = (1u << RG_FMT_SHIFT) - 1u;  // RgBlk::nchunks carries the format
constexpr uint32_t RG_PK_ACC_BITS = 13, RG_PK_ACC_MASK = (1u << RG_PK_ACC_BITS) - 1u;
__host__ __device__ constexpr uint32_t rg_fmt_word_bytes(uint32_t fmt) { return fmt == 2 ? 2u : fmt == 1 ? 3u : 4u; }
__host__ __device__ constexpr uint32_t rg_fmt_max_delta(uint32_t fmt) { return fmt == 2 ? 7u : 2047u; }
// byte offset of the value array inside a chunk / size of a chunk, by format (wide: the cold part and the short rows)
template <class T>
__host__ __device__ constexpr uint32_t rg_fmt_voff(bool wide, uint32_t fmt) {
  return wide ? 1536u : 256u * rg_fmt_word_bytes(fmt);
}
template <class T>
__host__ __device__ constexpr uint32_t rg_fmt_chunk_bytes(bool wide, uint32_t fmt) {
  return rg_fmt_voff<T>(wide, fmt) + 256u * (uint32_t)sizeof(T);
}
// inclusive prefix sum over the 64 lanes (DPP: Hillis-Steele inside the 16-lane rows, then the two row broadcasts)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t rg_dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t rg_wave_incl_scan(uint32_t v) {
  v += rg_dpp_u32<0x111, 0xF>(v);  // row_shr:1
  v += rg_dpp_u32<0x112, 0xF>(v);  // row_shr:2
  v += rg_dpp_u32<0x114, 0xF>(v);  // row_shr:4
  v += rg_dpp_u32<0x118, 0xF>(v);  // row_shr:8
  v += rg_dpp_u32<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
  v += rg_dpp_u32<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
  return v;
}
struct __attribute__((packed, aligned(4))) RgU3 {
  uint32_t x, y, z;
};
// position of the k-th entry of lane `lane` inside a chunk's value array
template <class T>
__host__ __device__ __forceinline__ uint32_t rg_val_slot(uint32_t lane, uint32_t k) {
  return sizeof(T) == 8 ? (k >> 1) * 128u + 2u * lane + (k & 1u) : 4u * lane + k;
}

// One (part, row group) block; blocks are stored part-major (b = part * ng + group), the short-row blocks behind them.
struct RgBlk {
  uint64_t off;      // byte offset of the block's first chunk in the stream
  uint64_t cp0;      // first pair of the block in cp (short-row blocks: first short row)
  uint32_t nchunks;  // 0: nothing to do
  uint32_t nheads;   // pairs of the block = head accumulators (short-row blocks: rows)
  uint32_t nsubs;    // extra accumulators behind the heads; bit 31 (queue entries only): a PIECE of a block, see RG_PIECE
  uint32_t sub0;     // their heads: sub_head[sub0 ..)
  uint64_t hdr;      // packed formats: byte offset of the block's chunk headers in the stream (one u32 per chunk: first slot)
};
// A block of a few rows and very many entries (a row longer than a whole group's share: its group holds little else) would
// keep one workgroup busy long after the others are done.  The kernel's queue therefore lists such a block as several
// PIECES — consecutive chunk ranges of the same block, same accumulators — whose sums are ADDED to the block's pairs by
// device-scope atomics (a handful of values per piece); the reduce kernel sets those pairs back to zero once it has read them.
constexpr uint32_t RG_PIECE = 0x80000000u;
constexpr uint32_t RG_PIECE_MAX_HEADS = 64;
static_assert(sizeof(RgBlk) == 40, "RgBlk layout");

// Tile t = rank slots [64 t, 64 t + 64), all of one group (a group starts on a multiple of 64: the slots skipped in
// front of it hold no row)
struct RgTile {
  uint32_t grp;
  uint16_t base[SPMV_NPART];  // pairs of the group's earlier tiles, per part
  uint16_t clear;             // (unused since round 4: the pairs that pieces add to are zeroed at the start of every product)
};
static_assert(sizeof(RgTile) == 24, "RgTile layout");

template <class T>
struct RgArgs {
  const RgBlk* blk;           // the queue entries: part-major, the short-row blocks behind them
  uint32_t qbase[SPMV_NPART + 1], qcount[SPMV_NPART + 1];  // entries [qbase[q], qbase[q] + qcount[q]) belong to queue q
  const uint32_t* sub_head;   // per extra accumulator: the head (block-local) it is folded into
  const unsigned char* stream;
  const T* base[SPMV_NPART];  // gather source of every part: the slab tables, x for the cold part
  T* cp;                      // the pair sums
  uint32_t* queue;            // NPART + 1 heads, SPMV_HEAD_STRIDE words apart (reset by sliced_build_tables_kernel)
  uint32_t abits;
  // short rows: a block of queue 9 holds short rows [cp0, cp0 + nheads), all their entries; y[short_rows[.]] directly
  T* y;
  const uint32_t* short_rows;
  unsigned long long* times;  // diagnostics (SPAM_SPMV_PROF=3): per block 4 wall-clock stamps (100 MHz); else null
};

template <class T>
__device__ __forceinline__ void rg_load_vals(const unsigned char* qv, int lane, T (&v)[4]) {
  if constexpr (sizeof(T) == 8) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    const V2 a = __builtin_nontemporal_load((const V2*)(qv + 16 * lane));
    const V2 b = __builtin_nontemporal_load((const V2*)(qv + 1024 + 16 * lane));
    v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
  } else {
    typedef T V4 __attribute__((ext_vector_type(4)));
    const V4 a = __builtin_nontemporal_load((const V4*)(qv + 16 * lane));
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
  }
}

// One block's chunks, dealt round-robin to the wavefronts.  A wavefront issues the stream loads of K chunks back to
// back, then their 4 K gathers, then the 4 K LDS adds: K * 3 KB per wavefront in flight and straight-line code between
// the waits (a cross-iteration software pipeline was measured no faster, and the compiler's register rotation made it
// wait for the loads it had just issued).
template <class T, int NT, int K, bool WIDE>
__device__ __forceinline__ void rg_stream(const unsigned char* p0, uint32_t nchunks, int wave, int lane,
                                          const T RG_GLB* xg, typename RgAcc<T>::type* acc, uint32_t abits) {
  typedef RgAcc<T> A;
  constexpr int nw = NT / 64;
  constexpr uint32_t CB = rg_chunk_bytes<T>(WIDE);
  constexpr uint32_t VOFF = WIDE ? 1536u : 1024u;
  const uint32_t amask = (1u << abits) - 1u;
  uint32_t c = (uint32_t)wave;
  for (; c + (uint32_t)(K - 1) * nw < nchunks; c += (uint32_t)K * nw) {
    u32x4 w[K];
    u32x2 a[K];
    T v[K][4], xv[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned char* q = p0 + (uint64_t)(c + (uint32_t)k * nw) * CB;
      w[k] = __builtin_nontemporal_load((const u32x4*)(q + 16 * lane));
      if (WIDE) a[k] = __builtin_nontemporal_load((const u32x2*)(q + 1024 + 8 * lane));
      rg_load_vals<T>(q + VOFF, lane, v[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (WIDE) {
        xv[k][0] = xg[w[k].x], xv[k][1] = xg[w[k].y], xv[k][2] = xg[w[k].z], xv[k][3] = xg[w[k].w];
      } else {
        xv[k][0] = xg[w[k].x >> abits], xv[k][1] = xg[w[k].y >> abits];
        xv[k][2] = xg[w[k].z >> abits], xv[k][3] = xg[w[k].w >> abits];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      uint32_t a0, a1, a2, a3;
      if (WIDE) {
        a0 = a[k].x & 0xFFFFu, a1 = a[k].x >> 16, a2 = a[k].y & 0xFFFFu, a3 = a[k].y >> 16;
      } else {
        a0 = w[k].x & amask, a1 = w[k].y & amask, a2 = w[k].z & amask, a3 = w[k].w & amask;
      }
      atomicAdd(&acc[a0], A::prod(v[k][0], xv[k][0]));
      atomicAdd(&acc[a1], A::prod(v[k][1], xv[k][1]));
      atomicAdd(&acc[a2], A::prod(v[k][2], xv[k][2]));
      atomicAdd(&acc[a3], A::prod(v[k][3], xv[k][3]));
    }
  }
  for (; c < nchunks; c += nw) {
    const unsigned char* q = p0 + (uint64_t)c * CB;
    const u32x4 w = __builtin_nontemporal_load((const u32x4*)(q + 16 * lane));
    u32x2 a = {0u, 0u};
    if (WIDE) a = __builtin_nontemporal_load((const u32x2*)(q + 1024 + 8 * lane));
    T v[4];
    rg_load_vals<T>(q + VOFF, lane, v);
    T x0, x1, x2, x3;
    uint32_t a0, a1, a2, a3;
    if (WIDE) {
      x0 = xg[w.x], x1 = xg[w.y], x2 = xg[w.z], x3 = xg[w.w];
      a0 = a.x & 0xFFFFu, a1 = a.x >> 16, a2 = a.y & 0xFFFFu, a3 = a.y >> 16;
    } else {
      x0 = xg[w.x >> abits], x1 = xg[w.y >> abits], x2 = xg[w.z >> abits], x3 = xg[w.w >> abits];
      a0 = w.x & amask, a1 = w.y & amask, a2 = w.z & amask, a3 = w.w & amask;
    }
    atomicAdd(&acc[a0], A::prod(v[0], x0));
    atomicAdd(&acc[a1], A::prod(v[1], x1));
    atomicAdd(&acc[a2], A::prod(v[2], x2));
    atomicAdd(&acc[a3], A::prod(v[3], x3));
  }
}

// The same for a block in a packed format (WB = 2 or 3 bytes per entry word): the slots of a chunk are its header's
// base plus the running sum of the delta fields in sorted order (entry 64 k + l = word k of lane l): one wave prefix sum
// per k, the total of each carried into the next.
template <class T, int NT, int K, int WB>
__device__ __forceinline__ void rg_stream_packed(const unsigned char* p0, const uint32_t* __restrict__ hdr, uint32_t nchunks,
                                                 int wave, int lane, const T RG_GLB* xg, typename RgAcc<T>::type* acc) {
  typedef RgAcc<T> A;
  constexpr int nw = NT / 64;
  constexpr uint32_t VOFF = 256u * (uint32_t)WB, CB = VOFF + 256u * (uint32_t)sizeof(T);
  auto load_words = [&](const unsigned char* q, uint32_t (&e)[4]) {
    if constexpr (WB == 2) {
      const u32x2 w = __builtin_nontemporal_load((const u32x2*)(q + 8 * lane));
      e[0] = w.x & 0xFFFFu, e[1] = w.x >> 16, e[2] = w.y & 0xFFFFu, e[3] = w.y >> 16;
    } else {
      const uint32_t* wp = (const uint32_t*)(q + 12 * lane);  // (three 4-byte nt loads: merged into one dwordx3)
      RgU3 w;
      w.x = __builtin_nontemporal_load(wp), w.y = __builtin_nontemporal_load(wp + 1), w.z = __builtin_nontemporal_load(wp + 2);
      e[0] = w.x & 0xFFFFFFu, e[1] = (w.x >> 24) | ((w.y & 0xFFFFu) << 8), e[2] = (w.y >> 16) | ((w.z & 0xFFu) << 16), e[3] = w.z >> 8;
    }
  };
  auto slots_of = [&](uint32_t base, const uint32_t (&e)[4], uint32_t (&sl)[4]) {
    uint32_t run = base;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t inc = rg_wave_incl_scan(e[j] >> RG_PK_ACC_BITS);
      sl[j] = run + inc;
      run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
  };
  uint32_t c = (uint32_t)wave;
  // (round 4 measured a software pipeline here — the index words and chunk headers of a batch loaded one batch ahead, the
  // values riding with the gathers, unconditional loads so that the compiler's vmcnt stays exact: 6 % SLOWER on both
  // bench matrices (0.617 against 0.579 ms; 0.226 against 0.213).  The batch still waits one full HBM round trip for its
  // values, the gathers it no longer waits for are L2 hits, and 133 live VGPRs against a budget of 128 spill.)
  for (; c + (uint32_t)(K - 1) * nw < nchunks; c += (uint32_t)K * nw) {
    uint32_t e[K][4], base[K], sl[K][4];
    T v[K][4], xv[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned char* q = p0 + (uint64_t)(c + (uint32_t)k * nw) * CB;
      base[k] = hdr[c + (uint32_t)k * nw];  // (uniform address)
      load_words(q, e[k]);
      rg_load_vals<T>(q + VOFF, lane, v[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) slots_of(base[k], e[k], sl[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      xv[k][0] = xg[sl[k][0]], xv[k][1] = xg[sl[k][1]], xv[k][2] = xg[sl[k][2]], xv[k][3] = xg[sl[k][3]];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&acc[e[k][j] & RG_PK_ACC_MASK], A::prod(v[k][j], xv[k][j]));
    }
  }
  for (; c < nchunks; c += nw) {
    const unsigned char* q = p0 + (uint64_t)c * CB;
    uint32_t e[4], sl[4];
    T v[4];
    const uint32_t base = hdr[c];
    load_words(q, e);
    rg_load_vals<T>(q + VOFF, lane, v);
    slots_of(base, e, sl);
    const T x0 = xg[sl[0]], x1 = xg[sl[1]], x2 = xg[sl[2]], x3 = xg[sl[3]];
    atomicAdd(&acc[e[0] & RG_PK_ACC_MASK], A::prod(v[0], x0));
    atomicAdd(&acc[e[1] & RG_PK_ACC_MASK], A::prod(v[1], x1));
    atomicAdd(&acc[e[2] & RG_PK_ACC_MASK], A::prod(v[2], x2));
    atomicAdd(&acc[e[3] & RG_PK_ACC_MASK], A::prod(v[3], x3));
  }
}

// Persistent workgroups.  A workgroup reads the XCD it runs on and serves that slab's queue, then the cold queue, the
// short rows, then whatever the other slabs have left (so nothing depends on the placement); one workgroup in four
// starts with the cold queue and the short rows, whose blocks wait for fabric round trips and overlap with the
// streaming of the others.
template <class T, int NT, int K>
__global__ __launch_bounds__(NT) void spmv_rg_kernel(RgArgs<T> P) {
  typedef typename RgAcc<T>::type AccT;
  extern __shared__ __align__(16) unsigned char rg_lds[];
  AccT* acc = reinterpret_cast<AccT*>(rg_lds);
  __shared__ uint32_t s_blk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t xcc = rg_xcc_id();
  const bool cold_first = ((blockIdx.x >> 3) & 3u) == 0u;
  int q = 0;
  for (;;) {
    uint32_t part;
    if (cold_first) part = q == 0 ? (uint32_t)SPMV_NSLAB : q == 1 ? RG_SHORT : q == 2 ? xcc : ((xcc + (uint32_t)(q - 2)) & 7u);
    else part = q == 0 ? xcc : q == 1 ? (uint32_t)SPMV_NSLAB : q == 2 ? RG_SHORT : ((xcc + (uint32_t)(q - 2)) & 7u);
    // (drawing the next ticket while the block streams is slower — 4 % with thread 0 drawing it in round 3, 7 % with the
    // last wavefront drawing it in round 4 (0.598 against 0.558 ms): a workgroup gets ~5 blocks per product, and committing
    // to one of them a whole block early costs more balance at the end than the ticket's round trip costs in between)
    if (threadIdx.x == 0) s_blk = atomicAdd(P.queue + (uint64_t)SPMV_HEAD_STRIDE * part, 1u);
    __syncthreads();
    const uint32_t g = __builtin_amdgcn_readfirstlane(s_blk);
    __syncthreads();
    const bool is_short = part == RG_SHORT;
    if (g >= P.qcount[part]) {
      if (++q > SPMV_NSLAB + 1) break;
      continue;
    }
    const RgBlk* bd = P.blk + (P.qbase[part] + g);
    const uint32_t nchunks = bd->nchunks & RG_NCHUNK_MASK, fmt = bd->nchunks >> RG_FMT_SHIFT;
    if (nchunks == 0 && !is_short) continue;  // no entry of this group in this part (its pairs, if any, stay zero); a
                                              // short-row block without entries still owes y its zeros
    const uint64_t off = bd->off, cp0 = bd->cp0;
    const uint32_t nheads = bd->nheads, nsubs = bd->nsubs & ~RG_PIECE, sub0 = bd->sub0;
    const bool piece = (bd->nsubs & RG_PIECE) != 0u;
    const uint32_t na = nheads + nsubs;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (P.times) t0 = wall_clock64();
    for (uint32_t i = threadIdx.x; i <= na; i += NT) acc[i] = AccT(0);  // [na] takes the padding entries
    __syncthreads();
    if (P.times) t1 = wall_clock64();
    if (part >= (uint32_t)SPMV_NSLAB)
      rg_stream<T, NT, K, true>(P.stream + off, nchunks, wave, lane, (const T RG_GLB*)P.base[SPMV_NSLAB], acc, P.abits);
    else if (fmt == 2)  // (uniform: a block has one format)
      rg_stream_packed<T, NT, K, 2>(P.stream + off, (const uint32_t*)(P.stream + bd->hdr), nchunks, wave, lane, (const T RG_GLB*)P.base[part], acc);
    else if (fmt == 1)
      rg_stream_packed<T, NT, K, 3>(P.stream + off, (const uint32_t*)(P.stream + bd->hdr), nchunks, wave, lane, (const T RG_GLB*)P.base[part], acc);
    else
      rg_stream<T, NT, K, false>(P.stream + off, nchunks, wave, lane, (const T RG_GLB*)P.base[part], acc, P.abits);
    __syncthreads();
    if (P.times) t2 = wall_clock64();
    if (nsubs) {  // uniform
      for (uint32_t i = threadIdx.x; i < nsubs; i += NT) atomicAdd(&acc[P.sub_head[sub0 + i]], acc[nheads + i]);
      __syncthreads();
    }
    if (is_short) {
      const uint32_t* sr = P.short_rows + cp0;
#pragma unroll 4
      for (uint32_t i = threadIdx.x; i < nheads; i += NT) P.y[sr[i]] = (T)acc[i];
    } else if (piece) {  // (uniform) the other pieces of the block may run on other XCDs: device scope
      T* out = P.cp + cp0;
      for (uint32_t i = threadIdx.x; i < nheads; i += NT)
        __hip_atomic_fetch_add(out + i, (T)acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      T* out = P.cp + cp0;
      for (uint32_t i = threadIdx.x; i < nheads; i += NT) out[i] = (T)acc[i];
    }
    __syncthreads();
    if (P.times && threadIdx.x == 0) {
      unsigned long long* t = P.times + 4ull * ((uint64_t)P.qbase[part] + g);
      t[0] = t0, t[1] = t1, t[2] = t2, t[3] = wall_clock64();
    }
  }
}

// y[row of rank slot] = sum of the row's pairs, in part order.  One wavefront per tile: a row's pair in part p is the
// (tile base + number of earlier slots of the tile with a pair in p)-th of block (p, group).  The tile's descriptor, its
// masks and its rows load side by side (the tile's first slot is 64 t), the pairs behind them: two dependent round trips
// per wavefront, at full occupancy.
// (Measured without gain: the tiles of the row-ordered ranks also zeroing the empty rows of their row range instead of a
// zero-fill of y; two or four tiles per wavefront — four cost the occupancy.)
template <class T>
__global__ __launch_bounds__(256) void spmv_rg_reduce_kernel(const RgTile* __restrict__ tiles, uint32_t ntiles,
                                                             const uint16_t* __restrict__ mask_slot,
                                                             const uint32_t* __restrict__ slot_rows,
                                                             const uint64_t* __restrict__ grp_cp, const T* __restrict__ cp,
                                                             T* __restrict__ y) {
  // (the tile index is wave-uniform: as a scalar, the descriptor and the nine block bases are scalar loads and leave the
  // vector-memory pipeline to the masks, rows, pairs and the store)
  const uint32_t t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)), lane = threadIdx.x & 63u;
  if (t >= ntiles) return;
  const uint32_t k = t * 64u + lane;
  const RgTile tl = tiles[t];
  const uint32_t m = mask_slot[k];  // 0 in a slot without row
  const uint32_t r = slot_rows[k];  // RG_NONE there
  const uint64_t lt = (1ull << lane) - 1ull;
  const uint64_t* gc = grp_cp + (uint64_t)tl.grp * SPMV_NPART;
  T v[SPMV_NPART];
#pragma unroll
  for (int p = 0; p < SPMV_NPART; ++p) {
    const bool has = (m >> p) & 1u;
    const uint64_t bal = __ballot(has);
    const uint64_t at = gc[p] + tl.base[p] + (uint64_t)__popcll(bal & lt);
    v[p] = has ? cp[at] : T(0);
  }
  T s = v[0];
#pragma unroll
  for (int p = 1; p < SPMV_NPART; ++p) s += v[p];
  if (r != RG_NONE) y[r] = s;
}

// ---- plan construction ------------------------------------------------------------------------------------------
// One wavefront per sliced row: length and the parts it has entries in.
__global__ void rg_row_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                              const uint8_t* __restrict__ slab, const uint32_t* __restrict__ long_rows, uint32_t nl,
                              uint32_t order_min, uint32_t* __restrict__ len, uint32_t* __restrict__ key,
                              uint32_t* __restrict__ idx, uint16_t* __restrict__ mask) {
  const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (w >= nl) return;
  const uint32_t r = long_rows[w];
  const uint32_t s = row_ptr[r], e = row_ptr[r + 1];
  uint32_t m = 0;
  for (uint32_t p = s + lane; p < e; p += 64) m |= 1u << slab[col[p]];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
  if (lane == 0) {
    len[w] = e - s;
    // ascending (stable) sort of ~len = descending lengths; rows below order_min entries all get one key and so keep
    // their row order behind the longer ones: the reduce then writes neighbouring y entries from neighbouring lanes
    key[w] = ~max(e - s, order_min - 1u);
    idx[w] = (uint32_t)w;
    mask[w] = (uint16_t)m;
  }
}
__global__ void rg_rank_kernel(const uint32_t* __restrict__ idx_of_rank, const uint16_t* __restrict__ mask,
                               const uint32_t* __restrict__ len, uint32_t nl, uint16_t* __restrict__ mask_rank,
                               uint32_t* __restrict__ len_rank) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nl) return;
  const uint32_t i = idx_of_rank[k];
  mask_rank[k] = mask[i];
  len_rank[k] = len[i];
}
// Ranks [k0, nl) in tiles of 64 (the ranks behind the hub rows are placed a whole tile at a time): per tile the number
// of rows with a pair in each part and the entries of its rows — what the host's group walk needs of 64 masks and lengths.
struct RgTileStat {
  uint8_t cnt[SPMV_NPART];
  uint8_t pad[3];
  uint32_t ents;
};
static_assert(sizeof(RgTileStat) == 16, "RgTileStat layout");
__global__ void rg_tile_stats_kernel(const uint16_t* __restrict__ mask_rank, const uint32_t* __restrict__ len_rank, uint32_t k0,
                                     uint32_t nl, uint32_t nt, RgTileStat* __restrict__ st) {
  const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (t >= nt) return;
  const uint32_t k = k0 + t * 64u + lane;
  const uint32_t m = k < nl ? mask_rank[k] : 0u;
  uint32_t e = k < nl ? len_rank[k] : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
  uint32_t mine = 0;
#pragma unroll
  for (int p = 0; p < SPMV_NPART; ++p) {
    const uint32_t c = (uint32_t)__popcll(__ballot((m >> p) & 1u));
    if ((int)lane == p) mine = c;
  }
  if (lane < (uint32_t)SPMV_NPART) st[t].cnt[lane] = (uint8_t)mine;
  if (lane == 0) st[t].ents = e;
}
// ... and their per-slot arrays: rank k0 + i sits in slot slot0 + i, in the group of its tile, with one accumulator per pair
__global__ void rg_slot_fill_kernel(const RgTile* __restrict__ tiles, const uint16_t* __restrict__ mask_rank, uint32_t k0,
                                    uint32_t slot0, uint32_t nl, uint32_t* __restrict__ rk_grp, uint32_t* __restrict__ rk_sub,
                                    uint16_t* __restrict__ mask_slot, uint32_t* __restrict__ slot_of_rank) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (k0 + i >= nl) return;
  const uint32_t sl = slot0 + i;
  rk_grp[sl] = tiles[sl >> 6].grp;
  rk_sub[sl] = 1u << 16;
  mask_slot[sl] = mask_rank[k0 + i];
  slot_of_rank[k0 + i] = sl;
}
// rank -> slot: the row of every slot and the slot of every sliced row
__global__ void rg_slot_kernel(const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ idx_of_rank,
                               const uint32_t* __restrict__ slot_of_rank, uint32_t nl, uint32_t* __restrict__ slot_rows,
                               uint32_t* __restrict__ slot_of) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nl) return;
  const uint32_t i = idx_of_rank[k], sl = slot_of_rank[k];
  slot_rows[sl] = long_rows[i];
  slot_of[i] = sl;
}
// One wavefront per tile: the head accumulator of every (slot, part) pair inside its block.
__global__ void rg_hidx_kernel(const RgTile* __restrict__ tiles, uint32_t ntiles, const uint16_t* __restrict__ mask_slot,
                               uint16_t* __restrict__ hidx) {
  const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (t >= ntiles) return;
  const RgTile tl = tiles[t];
  const uint32_t k = t * 64u + lane;
  const uint32_t m = mask_slot[k];
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int p = 0; p < SPMV_NPART; ++p) {
    const uint64_t bal = __ballot((m >> p) & 1u);
    hidx[(uint64_t)k * SPMV_NPART + p] = (uint16_t)(tl.base[p] + (uint32_t)__popcll(bal & lt));
  }
}

// One wavefront per sliced row: sort key (block | table slot or column | accumulator) and source position of every entry.
__global__ void rg_keys_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                               const uint32_t* __restrict__ code, const uint8_t* __restrict__ slab,
                               const uint32_t* __restrict__ long_rows, uint32_t nl, const uint64_t* __restrict__ lofs,
                               const uint32_t* __restrict__ rank_of, const uint32_t* __restrict__ rk_grp,
                               const uint32_t* __restrict__ rk_sub, const uint16_t* __restrict__ hidx,
                               const RgBlk* __restrict__ blk, uint32_t ng, uint64_t* __restrict__ keys,
                               uint32_t* __restrict__ src) {
  const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (w >= nl) return;
  const uint32_t r = long_rows[w];
  const uint32_t s = row_ptr[r], e = row_ptr[r + 1];
  const uint32_t k = rank_of[w];  // (its rank slot)
  const uint32_t g = rk_grp[k], rs = rk_sub[k];
  const uint32_t nsub = rs >> 16, sofs = rs & 0xFFFFu;  // accumulators of the row in a block: its head and nsub - 1 extra
  const uint64_t o = lofs[w];
  for (uint32_t p = s + lane; p < e; p += 64) {
    const uint32_t c = col[p];
    const uint32_t part = slab[c];
    const uint64_t b = (uint64_t)part * ng + g;
    const uint32_t sub = nsub > 1 ? (p - s) % nsub : 0u;
    const uint32_t acc = sub ? blk[b].nheads + sofs + sub - 1u : (uint32_t)hidx[(uint64_t)k * SPMV_NPART + part];
    keys[o + (p - s)] = (b << RG_KEY_BLK_SHIFT) | ((uint64_t)code[c] << RG_KEY_ACC_BITS) | (uint64_t)acc;
    src[o + (p - s)] = p;
  }
}

// One thread per short row (compact): its entries, unsliced, keyed by (short block, column).
__global__ void rg_short_keys_kernel(const uint32_t* __restrict__ res_ptr, const uint32_t* __restrict__ res_col, uint32_t ns,
                                     uint32_t srows, uint64_t first_blk, uint64_t out0, uint64_t* __restrict__ keys,
                                     uint32_t* __restrict__ src) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const uint64_t b = first_blk + i / srows;
  const uint32_t a = i % srows;
  for (uint32_t p = res_ptr[i], e = res_ptr[i + 1]; p < e; ++p) {
    keys[out0 + p] = (b << RG_KEY_BLK_SHIFT) | ((uint64_t)res_col[p] << RG_KEY_ACC_BITS) | (uint64_t)a;
    src[out0 + p] = p;
  }
}
__global__ void rg_fill_u64_kernel(uint64_t* __restrict__ p, uint64_t n, uint64_t v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void rg_bounds_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint64_t* __restrict__ start) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t b = keys[j] >> RG_KEY_BLK_SHIFT;
  if (j == 0 || (keys[j - 1] >> RG_KEY_BLK_SHIFT) != b) start[b] = j;
}

// block of a padded entry position: the last b with pe[b] <= q
__device__ __forceinline__ uint32_t rg_find_block(const uint64_t* __restrict__ pe, uint32_t nb, uint64_t q) {
  uint32_t lo = 0, hi = nb;  // pe[lo] <= q < pe[hi]
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (pe[mid] <= q) lo = mid;
    else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ uint32_t rg_key_slot(uint64_t key) { return (uint32_t)(key >> RG_KEY_ACC_BITS); }  // (low 32 bits of the shifted key)
// padding entries a packed format with deltas up to `maxd` needs in front of an entry that lies d slots after its predecessor
__device__ __forceinline__ uint32_t rg_pads(uint32_t d, uint32_t maxd) { return d ? (d - 1u) / maxd : 0u; }

// Per block of the slab parts: the padding entries the 16-bit and the 24-bit format would need (sum over its entries).
__global__ void rg_padsum_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint32_t first_wide,
                                 unsigned long long* __restrict__ sum16, unsigned long long* __restrict__ sum24) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t p16 = 0, p24 = 0;
  uint64_t b = ~0ull;
  if (j < n) {
    const uint64_t k = keys[j];
    b = k >> RG_KEY_BLK_SHIFT;
    if (b < first_wide && j > 0) {
      const uint64_t kp = keys[j - 1];
      if ((kp >> RG_KEY_BLK_SHIFT) == b) {
        const uint32_t d = rg_key_slot(k) - rg_key_slot(kp);
        p16 = rg_pads(d, rg_fmt_max_delta(2)), p24 = rg_pads(d, rg_fmt_max_delta(1));
      }
    }
  }
  // entries are sorted by block: a wavefront usually lies inside one
  const uint64_t b0 = __shfl(b, 0, 64);
  if (__ballot(b != b0) == 0ull) {
    uint32_t s16 = p16, s24 = p24;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s16 += __shfl_xor(s16, o, 64), s24 += __shfl_xor(s24, o, 64);
    if ((threadIdx.x & 63) == 0 && b0 != ~0ull && b0 < first_wide) {
      if (s16) atomicAdd(&sum16[b0], (unsigned long long)s16);
      if (s24) atomicAdd(&sum24[b0], (unsigned long long)s24);
    }
  } else if (b != ~0ull && b < first_wide) {
    if (p16) atomicAdd(&sum16[b], (unsigned long long)p16);
    if (p24) atomicAdd(&sum24[b], (unsigned long long)p24);
  }
}
// The padding in front of every entry under its block's chosen format (0 for fmt 0 and the wide blocks).
__global__ void rg_padsel_kernel(const uint64_t* __restrict__ keys, uint64_t n, const RgBlk* __restrict__ blk,
                                 uint32_t* __restrict__ pad) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t k = keys[j], b = k >> RG_KEY_BLK_SHIFT;
  const uint32_t fmt = blk[b].nchunks >> RG_FMT_SHIFT;
  uint32_t m = 0;
  if (fmt && j > 0) {
    const uint64_t kp = keys[j - 1];
    if ((kp >> RG_KEY_BLK_SHIFT) == b) m = rg_pads(rg_key_slot(k) - rg_key_slot(kp), rg_fmt_max_delta(fmt));
  }
  pad[j] = m;
}

// One thread per entry (in sorted order): write its word and value at the position the kernel's lanes read them from
// (sorted position inside the block, padding included, -> chunk, k, lane), the padding entries a packed format needs in
// front of it, and — the block's last entry — the padding that fills the block's last chunk.  The stream is zero and vsrc
// RG_NONE before: padding values and sources need no store.
// (first_wide = 8 * ng: the cold part and the short-row blocks use the wide chunk format; first_short = 9 * ng.)
template <class T>
__global__ void rg_fill_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ src_sorted,
                               const uint64_t* __restrict__ start, const uint64_t* __restrict__ pe,
                               const uint64_t* __restrict__ pad_before /* exclusive prefix of pad, n + 1; null: no packed block */,
                               const RgBlk* __restrict__ blk, uint32_t first_wide, uint32_t first_short,
                               uint32_t abits, const T* __restrict__ vals, const T* __restrict__ short_vals,
                               unsigned char* __restrict__ stream, uint32_t* __restrict__ vsrc, uint64_t n) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t key = keys[j];
  const uint32_t b = (uint32_t)(key >> RG_KEY_BLK_SHIFT);
  const uint32_t idx = rg_key_slot(key), a = (uint32_t)(key & ((1u << RG_KEY_ACC_BITS) - 1u));
  const bool is_short = b >= first_short, wide = b >= first_wide;
  const RgBlk bd = blk[b];
  const uint32_t nch = bd.nchunks & RG_NCHUNK_MASK, fmt = wide ? 0u : bd.nchunks >> RG_FMT_SHIFT;
  const uint32_t spare = bd.nheads + bd.nsubs;  // the block's spare accumulator takes the padding
  const uint64_t sb = start[b], last = start[b + 1] - 1;
  const uint32_t cb = wide ? rg_chunk_bytes<T>(true) : fmt == 2 ? rg_fmt_chunk_bytes<T>(false, 2)
                      : fmt == 1 ? rg_fmt_chunk_bytes<T>(false, 1) : rg_chunk_bytes<T>(false);
  const uint32_t voff = wide ? 1536u : fmt == 2 ? rg_fmt_voff<T>(false, 2) : fmt == 1 ? rg_fmt_voff<T>(false, 1) : 1024u;
  const uint32_t wb = rg_fmt_word_bytes(fmt);
  auto put = [&](uint64_t pos, uint32_t slot, uint32_t delta, uint32_t acc, bool real, uint32_t src, T v) {
    const uint64_t chunk = pos >> 8;
    const uint32_t w = (uint32_t)(pos & 255u), lane = w & 63u, k = w >> 6;
    unsigned char* c = stream + bd.off + chunk * cb;
    const uint32_t wpos = 4u * lane + k;
    if (wide) {
      reinterpret_cast<uint32_t*>(c)[wpos] = slot;
      reinterpret_cast<uint16_t*>(c + 1024)[wpos] = (uint16_t)acc;
    } else if (fmt == 0) {
      reinterpret_cast<uint32_t*>(c)[wpos] = (slot << abits) | acc;
    } else {
      if (w == 0) {  // the chunk's first entry: absolute slot in the block's header array, no delta
        reinterpret_cast<uint32_t*>(stream + bd.hdr)[chunk] = slot;
        delta = 0;
      }
      const uint32_t word = (delta << RG_PK_ACC_BITS) | acc;
      unsigned char* d = c + wb * wpos;
      if (fmt == 2) {
        *reinterpret_cast<uint16_t*>(d) = (uint16_t)word;
      } else {
        d[0] = (unsigned char)word, d[1] = (unsigned char)(word >> 8), d[2] = (unsigned char)(word >> 16);
      }
    }
    if (real) {
      const uint32_t vs = rg_val_slot<T>(lane, k);
      reinterpret_cast<T*>(c + voff)[vs] = v;
      if (vsrc) vsrc[pe[b] + (chunk << 8) + vs] = src;
    }
  };
  uint64_t pos = j - sb;
  uint32_t delta = 0;
  if (fmt) {
    const uint32_t m = (uint32_t)(pad_before[j + 1] - pad_before[j]);
    pos += pad_before[j + 1] - pad_before[sb];  // (the block's first entry has no padding in front of it)
    if (j > sb) {
      const uint32_t maxd = rg_fmt_max_delta(fmt), prev = rg_key_slot(keys[j - 1]);
      for (uint32_t i = 0; i < m; ++i) put(pos - m + i, prev + maxd * (i + 1u), maxd, spare, false, RG_NONE, T(0));
      delta = idx - (prev + maxd * m);
    }
  }
  put(pos, idx, delta, a, true, src_sorted[j], is_short ? short_vals[src_sorted[j]] : vals[src_sorted[j]]);
  if (j == last)
    for (uint64_t q = pos + 1; q < (uint64_t)nch * 256u; ++q) put(q, idx, 0u, spare, false, RG_NONE, T(0));
}

// spam_csr_values_changed: one thread per padded value position
template <class T>
__global__ void rg_refresh_kernel(const uint64_t* __restrict__ pe, const RgBlk* __restrict__ blk, uint32_t nb,
                                  uint32_t first_wide, uint32_t first_short, const uint32_t* __restrict__ vsrc,
                                  const T* __restrict__ vals, const T* __restrict__ short_vals,
                                  unsigned char* __restrict__ stream, uint64_t padded) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= padded) return;
  const uint32_t p = vsrc[q];
  if (p == RG_NONE) return;
  const uint32_t b = rg_find_block(pe, nb, q);
  const bool is_short = b >= first_short, wide = b >= first_wide;
  const uint32_t fmt = wide ? 0u : blk[b].nchunks >> RG_FMT_SHIFT;
  const uint32_t cb = fmt == 2 ? rg_fmt_chunk_bytes<T>(false, 2) : fmt == 1 ? rg_fmt_chunk_bytes<T>(false, 1) : rg_chunk_bytes<T>(wide);
  const uint32_t voff = fmt == 2 ? rg_fmt_voff<T>(false, 2) : fmt == 1 ? rg_fmt_voff<T>(false, 1) : (wide ? 1536u : 1024u);
  const uint64_t jb = q - pe[b];
  unsigned char* c = stream + blk[b].off + (jb >> 8) * cb;
  reinterpret_cast<T*>(c + voff)[jb & 255u] = is_short ? short_vals[p] : vals[p];
}

static int bits_for(uint64_t v) {  // bits needed to represent values 0..v
  int b = 1;
  while (b < 64 && (v >> b)) ++b;
  return b;
}

template <class T, int NT, int K>
static int rg_launch_or_grid(const RgPlan* g, const RgArgs<T>* args, hipStream_t s, unsigned* grid_out) {
  auto kern = spmv_rg_kernel<T, NT, K>;
  if (grid_out) {
    if (g->lds_bytes > 48 * 1024)
      SPAM_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds_bytes));
    int per_cu = 0, dev = 0, cus = 0;
    SPAM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, g->lds_bytes));
    SPAM_HIP_TRY(hipGetDevice(&dev));
    SPAM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (per_cu < 1) {  // (a device with less LDS than gfx950: declined, the caller keeps the tile form)
      *grid_out = 0;
      return SPAM_OK;
    }
    uint64_t n = (uint64_t)per_cu * (uint64_t)cus;
    if (n > g->nonempty_blocks) n = g->nonempty_blocks;
    *grid_out = (unsigned)std::max<uint64_t>(n, 1);
    return SPAM_OK;
  }
  kern<<<g->grid, NT, g->lds_bytes, s>>>(*args);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T>
static int rg_dispatch(const RgPlan* g, const RgArgs<T>* args, hipStream_t s, unsigned* grid_out) {
#define SPAM_RG_VARIANT(NT_, K_) \
  if (g->nt == NT_ && g->k == K_) return rg_launch_or_grid<T, NT_, K_>(g, args, s, grid_out)
  SPAM_RG_VARIANT(512, 4);
  SPAM_RG_VARIANT(1024, 2);
  SPAM_RG_VARIANT(1024, 4);
  SPAM_RG_VARIANT(256, 4);
#undef SPAM_RG_VARIANT
  if (grid_out) {  // an SPAM_SPMV_RG_NT / _K combination that is not compiled: declined like any other non-qualifying case
    *grid_out = 0;
    return SPAM_OK;
  }
  return set_error(SPAM_E_INVALID, "no row-group SpMV kernel for %d threads x %d chunks", g->nt, g->k);
}

struct RgGroup {
  uint32_t heads[SPMV_NPART];
  uint32_t nsubs;
  uint32_t first_slot;
  std::vector<uint32_t> sub_slot, sub_cnt;  // slots of the rows with extra accumulators, and how many
};

}  // namespace

int sort_u32_desc(const uint32_t* d_in, uint32_t* d_out, uint64_t n, hipStream_t s) {
  if (n == 0) return SPAM_OK;
  size_t tb = 0;
  SPAM_HIP_TRY(rocprim::radix_sort_keys_desc(nullptr, tb, d_in, d_out, (size_t)n, 0u, 32u, s));
  DevBuf tmp;
  SPAM_TRY(tmp.alloc(tb));
  SPAM_HIP_TRY(rocprim::radix_sort_keys_desc(tmp.p, tb, d_in, d_out, (size_t)n, 0u, 32u, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // tmp goes back to the pool
  return SPAM_OK;
}

int sort_u32_asc(const uint32_t* d_in, uint32_t* d_out, uint64_t n, int bits, hipStream_t s) {
  if (n == 0) return SPAM_OK;
  size_t tb = 0;
  SPAM_HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, d_in, d_out, (size_t)n, 0u, (unsigned)bits, s));
  DevBuf tmp;
  SPAM_TRY(tmp.alloc(tb));
  SPAM_HIP_TRY(rocprim::radix_sort_keys(tmp.p, tb, d_in, d_out, (size_t)n, 0u, (unsigned)bits, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // tmp goes back to the pool
  return SPAM_OK;
}

void rg_plan_free(RgPlan* g) {
  if (!g) return;
  void* ptrs[] = {g->stream, g->blk, g->qblk, g->pe, g->sub_head, g->slot_rows, g->mask_slot, g->tiles, g->grp_cp, g->cp, g->vsrc, g->piece_pairs};
  for (void* p : ptrs)
    if (p) dev_free(p);
  delete g;
}

template <class T>
int rg_build(const spam_csr* a, const uint8_t* slab, const uint32_t* code, const uint32_t* long_rows, uint32_t nl, uint32_t S,
             const spam_csr* res, uint32_t ns, const RgKnobs& kn, hipStream_t s, RgPlan** out) {
  *out = nullptr;
  if (nl == 0 || !a->row_ptr32) return SPAM_OK;
  // one word holds slot << abits | accumulator
  const int sbits = bits_for(S ? S - 1 : 0);
  int abits = std::min(13, 32 - sbits);
  if (kn.acc_bits > 0) abits = std::min(abits, kn.acc_bits);
  if (abits < 9) return SPAM_OK;  // tables too wide for the packed word: the caller keeps the tile form
  const uint32_t rmax = (1u << abits) - 1u;  // usable accumulators of a block (+ 1 spare for the padding entries)
  const uint64_t max_sub = std::min<uint64_t>(RG_MAX_SUB, rmax / 2);

  PhaseClock pc(kn.prof == 4, s);
  // -- sliced rows: length, part mask; ranks by length ----------------------------------------------------------------
  DevBuf len, key_in, key_out, idx_in, idx_out, mask, tmp;
  SPAM_TRY(len.alloc((uint64_t)nl * 4));
  SPAM_TRY(key_in.alloc((uint64_t)nl * 4));
  SPAM_TRY(key_out.alloc((uint64_t)nl * 4));
  SPAM_TRY(idx_in.alloc((uint64_t)nl * 4));
  SPAM_TRY(idx_out.alloc((uint64_t)nl * 4));
  SPAM_TRY(mask.alloc((uint64_t)nl * 2));
  const unsigned gl = (nl + 255) / 256;
  const unsigned gw = (unsigned)(((uint64_t)nl * 64 + 255) / 256);
  rg_row_kernel<<<gw, 256, 0, s>>>(a->row_ptr32, a->col_idx, slab, long_rows, nl, (uint32_t)std::max(1, kn.order_min),
                                   len.as<uint32_t>(), key_in.as<uint32_t>(), idx_in.as<uint32_t>(), mask.as<uint16_t>());
  SPAM_CHECK_LAUNCH();
  size_t tb = 0;
  SPAM_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, key_in.as<uint32_t>(), key_out.as<uint32_t>(), idx_in.as<uint32_t>(),
                                         idx_out.as<uint32_t>(), (size_t)nl, 0u, 32u, s));
  SPAM_TRY(tmp.alloc(tb));
  SPAM_HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, key_in.as<uint32_t>(), key_out.as<uint32_t>(), idx_in.as<uint32_t>(),
                                         idx_out.as<uint32_t>(), (size_t)nl, 0u, 32u, s));
  DevBuf mask_rank;
  SPAM_TRY(mask_rank.alloc((uint64_t)nl * 2 + 2));
  rg_rank_kernel<<<gl, 256, 0, s>>>(idx_out.as<uint32_t>(), mask.as<uint16_t>(), len.as<uint32_t>(), nl, mask_rank.as<uint16_t>(),
                                    key_in.as<uint32_t>() /* reused: lengths in rank order */);
  SPAM_CHECK_LAUNCH();
  // the host sees the lengths and masks of the leading ranks only (the hub rows, placed one by one below); the millions
  // of ranks behind them reach it as per-tile statistics (round 3 downloaded 6 bytes per sliced row and walked them all)
  uint32_t n_down = std::min<uint32_t>(nl, 1u << 16);
  std::vector<uint32_t> h_len(n_down);
  std::vector<uint16_t> h_mask(n_down);
  SPAM_HIP_TRY(hipMemcpyAsync(h_len.data(), key_in.p, (uint64_t)n_down * 4, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(h_mask.data(), mask_rank.p, (uint64_t)n_down * 2, hipMemcpyDeviceToHost, s));
  DevBuf lofs, stmp;
  SPAM_TRY(lofs.alloc(((uint64_t)nl + 1) * 8));
  SPAM_TRY(stmp.alloc(scan_tmp_bytes(nl)));
  SPAM_TRY(scan_u32_to_u64(len.as<uint32_t>(), nl, lofs.as<uint64_t>(), stmp.p, s));
  uint64_t n_ent = 0;
  SPAM_HIP_TRY(hipMemcpyAsync(&n_ent, lofs.as<uint64_t>() + nl, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (n_ent == 0 || n_ent >= 0xFFFFFFFFull) return SPAM_OK;
  pc.stamp("    rg: row lengths, masks, rank sort, downloads");
  // entries of a (part, row group) block.  200 k is the measured optimum when there are thousands of blocks (the bench
  // matrix: 100 k / 150 k / 200 k / 250 k -> 0.587 / 0.565 / 0.576 / 0.580 ms); smaller matrices need smaller blocks to
  // give every workgroup several (R-MAT scale 18 .. 21, 4 M .. 32 M entries, 200 k -> 16 k .. 50 k: 87 -> 29, 93 -> 42,
  // 111 -> 72, 148 -> 116 us; tools/size_sweep.py)
  const uint64_t n_all_est = n_ent + (res && kn.fold_short ? res->nnz : 0);
  const uint64_t eblk = kn.eblk > 0 ? (uint64_t)std::max(1024, kn.eblk)
                                    : std::min<uint64_t>(200000, std::max<uint64_t>(16000, n_all_est / 500));
  const uint64_t ecap = eblk * SPMV_NPART;  // entries of a group over its nine blocks

  // -- row groups and tiles (host): consecutive ranks; per part at most rmax accumulators, ecap entries in all.  Ranks
  //    are placed in SLOTS: a group starts on a multiple of 64 slots, so that tile t is slots [64 t, 64 t + 64) ----------
  std::vector<uint32_t> h_slot, h_sub_head;  // (h_slot and the by-slot vectors: the hub ranks' part only)
  std::vector<uint32_t> h_grp_s, h_rsub_s;   // by slot
  std::vector<uint16_t> h_mask_s;            // by slot (0 in a slot without row)
  std::vector<RgTile> h_tiles;
  std::vector<RgGroup> groups;
  uint32_t khub = 0, slot0 = 0;  // hub ranks; first slot of the ranks behind them
  {
    RgGroup cur{};
    uint64_t ents = 0;
    uint32_t slot = 0;
    auto open_tile = [&]() {
      RgTile t{};
      t.grp = (uint32_t)groups.size();
      for (int p = 0; p < SPMV_NPART; ++p) t.base[p] = (uint16_t)cur.heads[p];
      h_tiles.push_back(t);
      h_grp_s.resize((size_t)h_tiles.size() * 64, 0u), h_rsub_s.resize((size_t)h_tiles.size() * 64, 0u);
      h_mask_s.resize((size_t)h_tiles.size() * 64, (uint16_t)0);
    };
    open_tile();
    // The leading ranks whose rows take extra accumulators (nsub > 1: a few thousand rows at most) are placed one by one
    // and may close a group anywhere; everything behind them is placed a whole tile at a time and closes groups on tile
    // boundaries only (64 masks -> nine counts), which keeps this loop at a few milliseconds for millions of rows.
    auto nsub_of = [&](uint64_t L) {
      uint64_t n = (L * RG_MAX_SUB + ecap - 1) / ecap;
      return std::min<uint64_t>(std::max<uint64_t>(n, 1), std::min<uint64_t>(max_sub, std::max<uint64_t>(L, 1)));
    };
    // ranks [0, khub) are placed one by one (lengths are non-increasing there)
    for (;;) {
      while (khub < n_down && nsub_of(h_len[khub]) > 1) ++khub;
      if (khub < n_down || n_down == nl) break;
      // (more hub rows than downloaded so far: fetch twice as many)
      const uint32_t more = (uint32_t)std::min<uint64_t>(nl, (uint64_t)n_down * 2);
      h_len.resize(more), h_mask.resize(more);
      SPAM_HIP_TRY(hipMemcpyAsync(h_len.data() + n_down, (const uint32_t*)key_in.p + n_down, (uint64_t)(more - n_down) * 4, hipMemcpyDeviceToHost, s));
      SPAM_HIP_TRY(hipMemcpyAsync(h_mask.data() + n_down, (const uint16_t*)mask_rank.p + n_down, (uint64_t)(more - n_down) * 2, hipMemcpyDeviceToHost, s));
      SPAM_HIP_TRY(hipStreamSynchronize(s));
      n_down = more;
    }
    h_slot.resize(khub);
    for (uint32_t k = 0; k < khub; ++k) {
      const uint64_t L = h_len[k];
      // a row holding the fraction f of a block's entries draws 64 f lanes of every wave instruction: 8 x as many
      // accumulators keep two lanes off one LDS address most of the time
      const uint64_t nsub = nsub_of(L);
      if (nsub > 1) h_mask[k] = (uint16_t)RG_FULL_MASK;  // its extra accumulators exist in every part: so do its heads
      const uint32_t m = h_mask[k];
      uint32_t worst = 0;
      for (int p = 0; p < SPMV_NPART; ++p) worst = std::max(worst, cur.heads[p] + ((m >> p) & 1u));
      if (worst + cur.nsubs + (nsub - 1) > rmax || (ents > 0 && ents + L > ecap)) {
        groups.push_back(cur);
        cur = RgGroup{};
        ents = 0;
        slot = (slot + 63u) & ~63u;
        open_tile();
        cur.first_slot = slot;
      } else if (slot == (uint32_t)h_tiles.size() * 64u) {
        open_tile();
      }
      h_slot[k] = slot;
      h_grp_s[slot] = (uint32_t)groups.size();
      h_rsub_s[slot] = cur.nsubs | ((uint32_t)nsub << 16);
      h_mask_s[slot] = (uint16_t)m;
      if (nsub > 1) cur.sub_slot.push_back(slot), cur.sub_cnt.push_back((uint32_t)nsub - 1), cur.nsubs += (uint32_t)nsub - 1;
      for (int p = 0; p < SPMV_NPART; ++p) cur.heads[p] += (m >> p) & 1u;
      ents += L;
      ++slot;
    }
    if (khub < nl) {
      // the rest, tile by tile, from the next tile boundary on: the device condenses every 64 masks and lengths
      if (khub > 0) slot = (slot + 63u) & ~63u;
      slot0 = slot;
      const uint32_t rest_tiles = (uint32_t)(((uint64_t)(nl - khub) + 63) / 64);
      DevBuf d_st;
      SPAM_TRY(d_st.alloc((uint64_t)rest_tiles * sizeof(RgTileStat)));
      rg_tile_stats_kernel<<<(rest_tiles + 3) / 4, 256, 0, s>>>(mask_rank.as<uint16_t>(), key_in.as<uint32_t>(), khub, nl, rest_tiles,
                                                               d_st.as<RgTileStat>());
      SPAM_CHECK_LAUNCH();
      std::vector<RgTileStat> h_st(rest_tiles);
      SPAM_HIP_TRY(hipMemcpyAsync(h_st.data(), d_st.p, (uint64_t)rest_tiles * sizeof(RgTileStat), hipMemcpyDeviceToHost, s));
      SPAM_HIP_TRY(hipStreamSynchronize(s));
      h_tiles.reserve(h_tiles.size() + rest_tiles);
      if (slot < (uint32_t)h_tiles.size() * 64u) h_tiles.pop_back();  // (the tile opened before the loop, still without rows)
      for (uint32_t ti = 0; ti < rest_tiles; ++ti, slot += 64) {
        const RgTileStat& ts = h_st[ti];
        const uint64_t tents = ts.ents;
        uint32_t worst = 0;
        for (int p = 0; p < SPMV_NPART; ++p) worst = std::max(worst, cur.heads[p] + (uint32_t)ts.cnt[p]);
        const bool group_has_rows = ents > 0;
        if (group_has_rows && (worst + cur.nsubs > rmax || ents + tents > ecap)) {
          groups.push_back(cur);
          cur = RgGroup{};
          ents = 0;
          cur.first_slot = slot;
        }
        RgTile t{};
        t.grp = (uint32_t)groups.size();
        for (int p = 0; p < SPMV_NPART; ++p) t.base[p] = (uint16_t)cur.heads[p];
        h_tiles.push_back(t);
        for (int p = 0; p < SPMV_NPART; ++p) cur.heads[p] += ts.cnt[p];
        ents += tents;
      }
    }
    groups.push_back(cur);
  }
  pc.stamp("    rg: groups and tiles on the host");
  const uint32_t ng = (uint32_t)groups.size();
  const uint32_t ntiles = (uint32_t)h_tiles.size();
  const uint64_t nslots = (uint64_t)ntiles * 64;
  // short rows (the residual CSR, compacted by the caller): groups of rmax consecutive rows behind the nine parts
  const bool fold = kn.fold_short && res && ns > 0 && res->nnz > 0;
  const uint32_t srows = rmax;
  const uint32_t ngs = fold ? (ns + srows - 1) / srows : 0;
  const uint64_t n_short = fold ? res->nnz : 0;
  if (n_ent + n_short >= 0xFFFFFFFFull) return SPAM_OK;
  const uint64_t nb64 = (uint64_t)ng * SPMV_NPART + ngs;
  const int bbits = bits_for(nb64 - 1);
  if (RG_KEY_BLK_SHIFT + bbits > 64) return SPAM_OK;
  const uint32_t nb = (uint32_t)nb64, first_wide = ng * SPMV_NSLAB, first_short = ng * SPMV_NPART;

  // block descriptors (stream offsets follow once the entries are counted); pairs in (group, part) order
  std::vector<RgBlk> h_blk((size_t)nb + 1, RgBlk{});
  std::vector<uint64_t> h_grp_cp((size_t)ng * SPMV_NPART);
  uint64_t npairs = 0;
  for (uint32_t g = 0; g < ng; ++g)
    for (int p = 0; p < SPMV_NPART; ++p) {
      RgBlk& b = h_blk[(size_t)p * ng + g];
      b.nheads = groups[g].heads[p], b.nsubs = groups[g].nsubs, b.cp0 = npairs;
      h_grp_cp[(size_t)g * SPMV_NPART + p] = npairs;
      npairs += b.nheads;
    }
  for (uint32_t gs = 0; gs < ngs; ++gs) {
    RgBlk& b = h_blk[(size_t)first_short + gs];
    b.cp0 = (uint64_t)gs * srows, b.nheads = std::min(srows, ns - gs * srows);
  }
  // extra accumulators: per block, the head of each inside that block (a rank's head index differs from part to part)
  for (uint32_t gi = 0; gi < ng; ++gi) {
    const RgGroup& G = groups[gi];
    if (G.nsubs == 0) continue;
    for (int p = 0; p < SPMV_NPART; ++p) {
      h_blk[(size_t)p * ng + gi].sub0 = (uint32_t)h_sub_head.size();
      size_t si = 0;
      uint32_t c = 0;  // pairs of part p among the group's ranks before k (the rule of rg_hidx_kernel)
      for (uint32_t sl = G.first_slot; si < G.sub_slot.size(); ++sl) {
        if (sl == G.sub_slot[si]) {
          for (uint32_t j = 0; j < G.sub_cnt[si]; ++j) h_sub_head.push_back(c);
          ++si;
        }
        c += (h_mask_s[sl] >> p) & 1u;
      }
    }
  }

  RgPlan* g = new (std::nothrow) RgPlan();
  if (!g) return set_error(SPAM_E_OOM, "host allocation failed");
  struct Guard {
    RgPlan* g;
    ~Guard() { if (g) rg_plan_free(g); }
  } guard{g};
  g->ng = ng, g->abits = (uint32_t)abits, g->entries = n_ent + n_short, g->npairs = npairs, g->ntiles = ntiles, g->nl = nl;
  g->ngs = ngs, g->ns = fold ? ns : 0;
  g->nt = kn.nt, g->k = kn.k, g->prof = kn.prof;
  g->lds_bytes = (rmax + 1u) * 8u;
  g->naccs = npairs + h_sub_head.size();

  DevBuf rk_grp, rk_sub, tiles, hidx, blk, slot_of_rank, slot_rows, slot_of, mask_slot;
  SPAM_TRY(rk_grp.alloc(nslots * 4));
  SPAM_TRY(rk_sub.alloc(nslots * 4));
  SPAM_TRY(tiles.alloc((uint64_t)ntiles * sizeof(RgTile)));
  SPAM_TRY(hidx.alloc(nslots * SPMV_NPART * 2));
  SPAM_TRY(blk.alloc(((uint64_t)nb + 1) * sizeof(RgBlk)));
  SPAM_TRY(slot_of_rank.alloc((uint64_t)nl * 4));
  SPAM_TRY(slot_rows.alloc(nslots * 4));
  SPAM_TRY(slot_of.alloc((uint64_t)nl * 4));
  SPAM_TRY(mask_slot.alloc(nslots * 2));
  SPAM_HIP_TRY(hipMemsetAsync(rk_grp.p, 0, nslots * 4, s));
  SPAM_HIP_TRY(hipMemsetAsync(rk_sub.p, 0, nslots * 4, s));
  SPAM_HIP_TRY(hipMemsetAsync(mask_slot.p, 0, nslots * 2, s));  // 0 in a slot without row
  SPAM_HIP_TRY(hipMemcpyAsync(tiles.p, h_tiles.data(), (uint64_t)ntiles * sizeof(RgTile), hipMemcpyHostToDevice, s));
  {  // the hub ranks' slots from the host (with the forced masks), everything behind them on the device
    const uint64_t hub_slots = std::min<uint64_t>(h_grp_s.size(), nslots);
    if (hub_slots) {
      SPAM_HIP_TRY(hipMemcpyAsync(rk_grp.p, h_grp_s.data(), hub_slots * 4, hipMemcpyHostToDevice, s));
      SPAM_HIP_TRY(hipMemcpyAsync(rk_sub.p, h_rsub_s.data(), hub_slots * 4, hipMemcpyHostToDevice, s));
      SPAM_HIP_TRY(hipMemcpyAsync(mask_slot.p, h_mask_s.data(), hub_slots * 2, hipMemcpyHostToDevice, s));
    }
    if (khub) SPAM_HIP_TRY(hipMemcpyAsync(slot_of_rank.p, h_slot.data(), (uint64_t)khub * 4, hipMemcpyHostToDevice, s));
    if (khub < nl) {
      rg_slot_fill_kernel<<<(nl - khub + 255) / 256, 256, 0, s>>>(tiles.as<RgTile>(), mask_rank.as<uint16_t>(), khub, slot0, nl,
                                                                rk_grp.as<uint32_t>(), rk_sub.as<uint32_t>(), mask_slot.as<uint16_t>(),
                                                                slot_of_rank.as<uint32_t>());
      SPAM_CHECK_LAUNCH();
    }
  }
  SPAM_HIP_TRY(hipMemcpyAsync(blk.p, h_blk.data(), ((uint64_t)nb + 1) * sizeof(RgBlk), hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemsetAsync(slot_rows.p, 0xFF, nslots * 4, s));  // RG_NONE in the slots without row
  rg_slot_kernel<<<gl, 256, 0, s>>>(long_rows, idx_out.as<uint32_t>(), slot_of_rank.as<uint32_t>(), nl, slot_rows.as<uint32_t>(),
                                    slot_of.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  rg_hidx_kernel<<<(ntiles + 3) / 4, 256, 0, s>>>(tiles.as<RgTile>(), ntiles, mask_slot.as<uint16_t>(), hidx.as<uint16_t>());
  SPAM_CHECK_LAUNCH();

  pc.stamp("    rg: descriptors, uploads, slot kernels");
  // -- entries: key, sort by (block, slot) -------------------------------------------------------------------------------
  DevBuf keys_in, keys_out, src_in, src_out;
  const uint64_t n_all = n_ent + n_short;
  SPAM_TRY(keys_in.alloc(n_all * 8));
  SPAM_TRY(keys_out.alloc(n_all * 8));
  SPAM_TRY(src_in.alloc(n_all * 4));
  SPAM_TRY(src_out.alloc(n_all * 4));
  rg_keys_kernel<<<gw, 256, 0, s>>>(a->row_ptr32, a->col_idx, code, slab, long_rows, nl, lofs.as<uint64_t>(), slot_of.as<uint32_t>(),
                                    rk_grp.as<uint32_t>(), rk_sub.as<uint32_t>(), hidx.as<uint16_t>(), blk.as<RgBlk>(), ng,
                                    keys_in.as<uint64_t>(), src_in.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  if (fold) {
    rg_short_keys_kernel<<<(ns + 255) / 256, 256, 0, s>>>(res->row_ptr32, res->col_idx, ns, srows, (uint64_t)first_short, n_ent,
                                                         keys_in.as<uint64_t>(), src_in.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
  }
  const int end_bit = RG_KEY_BLK_SHIFT + bbits;
  tb = 0;
  // (the two key / source buffers are the sort's ping and pong: no third copy of 162.7 M twelve-byte pairs in temporary storage)
  rocprim::double_buffer<uint64_t> keys_db(keys_in.as<uint64_t>(), keys_out.as<uint64_t>());
  rocprim::double_buffer<uint32_t> src_db(src_in.as<uint32_t>(), src_out.as<uint32_t>());
  SPAM_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, keys_db, src_db, (size_t)n_all, (unsigned)RG_KEY_ACC_BITS, (unsigned)end_bit, s));
  SPAM_TRY(tmp.alloc(tb));
  SPAM_HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, keys_db, src_db, (size_t)n_all, (unsigned)RG_KEY_ACC_BITS, (unsigned)end_bit, s));
  if (keys_db.current() != keys_out.as<uint64_t>()) std::swap(keys_in.p, keys_out.p);  // the sorted pairs are in keys_out / src_out
  if (src_db.current() != src_out.as<uint32_t>()) std::swap(src_in.p, src_out.p);
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // the host vectors above are read by the copies
  keys_in.reset(), src_in.reset(), tmp.reset(), hidx.reset();

  pc.stamp("    rg: entry keys + radix sort (incl. allocations)");
  // -- block bounds, chunk offsets ---------------------------------------------------------------------------------------
  DevBuf start;
  SPAM_TRY(start.alloc(((uint64_t)nb + 1) * 8));
  rg_fill_u64_kernel<<<(nb + 1 + 255) / 256, 256, 0, s>>>(start.as<uint64_t>(), (uint64_t)nb + 1, ~0ull);
  SPAM_CHECK_LAUNCH();
  rg_bounds_kernel<<<(unsigned)((n_all + 255) / 256), 256, 0, s>>>(keys_out.as<uint64_t>(), n_all, start.as<uint64_t>());
  SPAM_CHECK_LAUNCH();
  std::vector<uint64_t> h_start((size_t)nb + 1), h_pe((size_t)nb + 1);
  // the packed formats: padding each would need, per block of the slab parts
  const int fmt_max = abits <= (int)RG_PK_ACC_BITS ? std::max(0, std::min(2, kn.fmt)) : 0;
  DevBuf sum16, sum24;
  std::vector<unsigned long long> h16((size_t)first_wide + 1, 0ull), h24((size_t)first_wide + 1, 0ull);
  if (fmt_max > 0) {
    SPAM_TRY(sum16.alloc(((uint64_t)first_wide + 1) * 8));
    SPAM_TRY(sum24.alloc(((uint64_t)first_wide + 1) * 8));
    SPAM_HIP_TRY(hipMemsetAsync(sum16.p, 0, ((uint64_t)first_wide + 1) * 8, s));
    SPAM_HIP_TRY(hipMemsetAsync(sum24.p, 0, ((uint64_t)first_wide + 1) * 8, s));
    rg_padsum_kernel<<<(unsigned)((n_all + 255) / 256), 256, 0, s>>>(keys_out.as<uint64_t>(), n_all, first_wide,
                                                                     sum16.as<unsigned long long>(), sum24.as<unsigned long long>());
    SPAM_CHECK_LAUNCH();
    SPAM_HIP_TRY(hipMemcpyAsync(h16.data(), sum16.p, (uint64_t)first_wide * 8, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipMemcpyAsync(h24.data(), sum24.p, (uint64_t)first_wide * 8, hipMemcpyDeviceToHost, s));
  }
  SPAM_HIP_TRY(hipMemcpyAsync(h_start.data(), start.p, ((uint64_t)nb + 1) * 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  h_start[nb] = n_all;
  for (uint32_t b = nb; b-- > 0;)
    if (h_start[b] == ~0ull) h_start[b] = h_start[b + 1];
  h_pe[0] = 0;
  uint64_t off = 0, nonempty = 0, fmt_entries[3] = {0, 0, 0}, fmt_pads = 0;
  bool any_packed = false;
  for (uint32_t b = 0; b < nb; ++b) {
    const uint64_t cnt = h_start[b + 1] - h_start[b];
    const bool wide = b >= first_wide;
    // the format with the fewest bytes (the plain one for the cold part, the short rows and sparse blocks)
    uint32_t fmt = 0;
    uint64_t nel = cnt, bytes = ((cnt + 255) / 256) * rg_chunk_bytes<T>(wide);
    if (!wide && cnt > 0 && fmt_max > 0) {
      const uint64_t n24 = cnt + h24[b], c24 = (n24 + 255) / 256, b24 = c24 * rg_fmt_chunk_bytes<T>(false, 1) + ((c24 * 4 + 127) / 128) * 128;
      if (b24 < bytes && n24 < (1ull << 38)) fmt = 1, nel = n24, bytes = c24 * rg_fmt_chunk_bytes<T>(false, 1);
      if (fmt_max > 1) {
        const uint64_t n16 = cnt + h16[b], c16 = (n16 + 255) / 256, b16 = c16 * rg_fmt_chunk_bytes<T>(false, 2) + ((c16 * 4 + 127) / 128) * 128;
        if (b16 < bytes + (fmt ? ((nel + 255) / 256 * 4 + 127) / 128 * 128 : 0) && n16 < (1ull << 38)) fmt = 2, nel = n16, bytes = c16 * rg_fmt_chunk_bytes<T>(false, 2);
      }
    }
    const uint64_t nch = (nel + 255) / 256;
    if (nch > RG_NCHUNK_MASK) return SPAM_OK;  // (a block of 2^38 entries: the caller keeps the tile form)
    const uint64_t hdr_bytes = fmt ? ((nch * 4 + 127) / 128) * 128 : 0;
    h_blk[b].hdr = off, h_blk[b].off = off + hdr_bytes, h_blk[b].nchunks = (uint32_t)nch | (fmt << RG_FMT_SHIFT);
    off += hdr_bytes + bytes;
    h_pe[b + 1] = h_pe[b] + nch * 256;
    nonempty += cnt != 0;
    fmt_entries[fmt] += cnt, fmt_pads += nel - cnt;
    any_packed |= fmt != 0;
  }
  if (kn.prof == 4)
    fprintf(stderr, "[spam] prepare:     rg: entry words 32 / 24 / 16 bit: %llu / %llu / %llu entries, %llu padding entries, stream %.3f GB\n",
            (unsigned long long)fmt_entries[0], (unsigned long long)fmt_entries[1], (unsigned long long)fmt_entries[2],
            (unsigned long long)fmt_pads, (double)off / 1e9);
  h_blk[nb].off = off;
  g->stream_bytes = off, g->padded = h_pe[nb];
  // the kernel's queues: the blocks part by part, a block of few rows and many entries as pieces (RG_PIECE)
  std::vector<RgBlk> h_q;
  h_q.reserve((size_t)nb + 64);
  std::vector<char> grp_clear(ng, 0);
  std::vector<uint64_t> h_piece_pairs;  // pairs that pieces ADD to: zeroed at the start of every product (sliced_build_tables)
  {
    const uint32_t piece_chunks = (uint32_t)std::max<uint64_t>(1, eblk / 256);
    for (uint32_t part = 0; part <= (uint32_t)SPMV_NPART; ++part) {
      g->qbase[part] = (uint32_t)h_q.size();
      const uint32_t cnt = part < (uint32_t)SPMV_NPART ? ng : ngs;
      for (uint32_t i = 0; i < cnt; ++i) {
        const RgBlk d = h_blk[(size_t)part * ng + i];
        const uint32_t d_nch = d.nchunks & RG_NCHUNK_MASK, d_fmt = d.nchunks >> RG_FMT_SHIFT;
        if (part < (uint32_t)SPMV_NPART && d.nheads > 0 && d.nheads <= RG_PIECE_MAX_HEADS && d_nch > 2 * piece_chunks) {
          const uint32_t m = (d_nch + piece_chunks - 1) / piece_chunks, per = (d_nch + m - 1) / m;
          const uint64_t cb = rg_fmt_chunk_bytes<T>(part >= (uint32_t)SPMV_NSLAB, d_fmt);
          for (uint32_t c0 = 0; c0 < d_nch; c0 += per) {
            RgBlk e = d;
            e.off = d.off + (uint64_t)c0 * cb, e.hdr = d.hdr + (uint64_t)c0 * 4, e.nchunks = std::min(per, d_nch - c0) | (d_fmt << RG_FMT_SHIFT), e.nsubs = d.nsubs | RG_PIECE;
            h_q.push_back(e);
          }
          grp_clear[i] = 1;
          for (uint32_t hh = 0; hh < d.nheads; ++hh) h_piece_pairs.push_back(d.cp0 + hh);
        } else {
          h_q.push_back(d);
        }
      }
      g->qcount[part] = (uint32_t)h_q.size() - g->qbase[part];
    }
    for (RgTile& t : h_tiles) t.clear = 0;  // (round 3: the reduce set those pairs back to zero — a product then depended on
                                            // its predecessor's epilogue having run; now every product zeroes them first)
  }
  g->nq = (uint32_t)h_q.size();
  nonempty = 0;
  for (const RgBlk& e : h_q) nonempty += (e.nchunks & RG_NCHUNK_MASK) != 0;
  g->nonempty_blocks = nonempty;
  DevBuf pe, stream, vsrc, sub_head, grp_cp, cp, qblk;
  DevBuf piece_pairs;
  SPAM_TRY(piece_pairs.alloc((h_piece_pairs.size() + 1) * 8));
  if (!h_piece_pairs.empty())
    SPAM_HIP_TRY(hipMemcpyAsync(piece_pairs.p, h_piece_pairs.data(), h_piece_pairs.size() * 8, hipMemcpyHostToDevice, s));
  g->n_piece_pairs = (uint32_t)h_piece_pairs.size();
  SPAM_TRY(qblk.alloc(((uint64_t)h_q.size() + 1) * sizeof(RgBlk)));
  SPAM_HIP_TRY(hipMemcpyAsync(qblk.p, h_q.data(), h_q.size() * sizeof(RgBlk), hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(tiles.p, h_tiles.data(), (uint64_t)ntiles * sizeof(RgTile), hipMemcpyHostToDevice, s));  // (with the clear flags)
  SPAM_TRY(pe.alloc(((uint64_t)nb + 1) * 8));
  SPAM_TRY(stream.alloc(g->stream_bytes + 16));
  if (kn.keep_vsrc) SPAM_TRY(vsrc.alloc(g->padded * 4 + 16));
  SPAM_TRY(sub_head.alloc(h_sub_head.size() * 4 + 4));
  SPAM_TRY(grp_cp.alloc((uint64_t)ng * SPMV_NPART * 8));
  SPAM_TRY(cp.alloc((npairs + 2) * sizeof(T)));
  SPAM_HIP_TRY(hipMemsetAsync(cp.p, 0, (npairs + 2) * sizeof(T), s));  // pairs without entries (forced masks) stay zero
  SPAM_HIP_TRY(hipMemcpyAsync(start.p, h_start.data(), ((uint64_t)nb + 1) * 8, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(blk.p, h_blk.data(), ((uint64_t)nb + 1) * sizeof(RgBlk), hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(pe.p, h_pe.data(), ((uint64_t)nb + 1) * 8, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(grp_cp.p, h_grp_cp.data(), (uint64_t)ng * SPMV_NPART * 8, hipMemcpyHostToDevice, s));
  if (!h_sub_head.empty())
    SPAM_HIP_TRY(hipMemcpyAsync(sub_head.p, h_sub_head.data(), h_sub_head.size() * 4, hipMemcpyHostToDevice, s));
  // padding in front of every entry under its block's format, prefix-summed: an entry's position inside its block
  DevBuf padv, pad_before, ptmp;
  if (any_packed) {
    SPAM_TRY(padv.alloc(n_all * 4));
    SPAM_TRY(pad_before.alloc((n_all + 1) * 8));
    SPAM_TRY(ptmp.alloc(scan_tmp_bytes(n_all)));
    rg_padsel_kernel<<<(unsigned)((n_all + 255) / 256), 256, 0, s>>>(keys_out.as<uint64_t>(), n_all, blk.as<RgBlk>(), padv.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
    SPAM_TRY(scan_u32_to_u64(padv.as<uint32_t>(), n_all, pad_before.as<uint64_t>(), ptmp.p, s));
  }
  SPAM_HIP_TRY(hipMemsetAsync(stream.p, 0, g->stream_bytes + 16, s));     // padding values are zero
  if (vsrc.p) SPAM_HIP_TRY(hipMemsetAsync(vsrc.p, 0xFF, g->padded * 4 + 16, s));  // ... and have no source (RG_NONE)
  rg_fill_kernel<T><<<(unsigned)((n_all + 255) / 256), 256, 0, s>>>(
      keys_out.as<uint64_t>(), src_out.as<uint32_t>(), start.as<uint64_t>(), pe.as<uint64_t>(),
      any_packed ? pad_before.as<uint64_t>() : nullptr, blk.as<RgBlk>(), first_wide, first_short, (uint32_t)abits,
      (const T*)a->vals, fold ? (const T*)res->vals : nullptr, stream.as<unsigned char>(), vsrc.as<uint32_t>(), n_all);
  SPAM_CHECK_LAUNCH();
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // host vectors and temporaries go out of scope
  pc.stamp("    rg: bounds, offsets, stream fill (incl. allocations)");
  g->stream = stream.as<unsigned char>(), stream.release();
  g->blk = blk.release();
  g->qblk = qblk.release();
  g->pe = pe.as<uint64_t>(), pe.release();
  g->sub_head = sub_head.as<uint32_t>(), sub_head.release();
  g->slot_rows = slot_rows.as<uint32_t>(), slot_rows.release();
  g->mask_slot = mask_slot.as<uint16_t>(), mask_slot.release();
  g->tiles = tiles.release();
  g->grp_cp = grp_cp.as<uint64_t>(), grp_cp.release();
  g->cp = cp.release();
  g->vsrc = vsrc.p ? vsrc.as<uint32_t>() : nullptr, vsrc.release();
  g->piece_pairs = piece_pairs.as<uint64_t>(), piece_pairs.release();
  RgArgs<T> none{};
  SPAM_TRY(rg_dispatch<T>(g, &none, s, &g->grid));
  if (g->grid == 0) return SPAM_OK;  // no kernel for these knobs / this device: *out stays null, the guard frees the plan
  guard.g = nullptr;
  *out = g;
  return SPAM_OK;
}

template <class T>
int rg_run(const RgPlan* g, const T* xh, uint32_t S, const T* x, uint32_t* queue, T* y, const uint32_t* short_rows,
           hipStream_t s) {
  RgArgs<T> P;
  P.blk = (const RgBlk*)g->qblk, P.sub_head = g->sub_head, P.stream = g->stream;
  for (int q = 0; q <= SPMV_NPART; ++q) P.qbase[q] = g->qbase[q], P.qcount[q] = g->qcount[q];
  for (int p = 0; p < SPMV_NSLAB; ++p) P.base[p] = xh + (uint64_t)p * S;
  P.base[SPMV_NSLAB] = x;
  P.cp = (T*)g->cp, P.queue = queue, P.abits = g->abits;
  P.y = y, P.short_rows = short_rows;
  P.times = nullptr;
  DevBuf tbuf;
  const uint64_t nblk = g->nq;
  if (g->prof == 3) {  // diagnostics: where the workgroups' time goes (one synchronous launch, stderr)
    SPAM_TRY(tbuf.alloc(nblk * 32));
    SPAM_HIP_TRY(hipMemsetAsync(tbuf.p, 0, nblk * 32, s));
    P.times = tbuf.as<unsigned long long>();
  }
  SPAM_TRY(rg_dispatch<T>(g, &P, s, nullptr));
  if (P.times) {
    std::vector<unsigned long long> h(nblk * 4);
    SPAM_HIP_TRY(hipMemcpyAsync(h.data(), tbuf.p, nblk * 32, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    unsigned long long lo = ~0ull, hi = 0;
    double zero = 0, stream = 0, write = 0;
    uint64_t ran = 0;
    for (uint64_t b = 0; b < nblk; ++b) {
      const unsigned long long* t = &h[4 * b];
      if (!t[3]) continue;
      lo = std::min(lo, t[0]), hi = std::max(hi, t[3]);
      zero += (double)(t[1] - t[0]), stream += (double)(t[2] - t[1]), write += (double)(t[3] - t[2]);
      ++ran;
    }
    const double span = (double)(hi - lo), cap = span * g->grid;
    fprintf(stderr, "[spam] row-group SpMV kernel: %llu blocks on %u workgroups, span %.1f us; workgroup time: zero-fill %.1f %%, "
            "stream %.1f %%, fold + store %.1f %%, between blocks (tickets, descriptors, idle at the end) %.1f %%\n",
            (unsigned long long)ran, g->grid, span / 100.0, 100 * zero / cap, 100 * stream / cap, 100 * write / cap,
            100 * (cap - zero - stream - write) / cap);
    for (int p = 0; p <= SPMV_NPART; ++p) {
      const uint64_t b0 = g->qbase[p], b1 = b0 + g->qcount[p];
      unsigned long long last = 0;
      double st = 0;
      for (uint64_t b = b0; b < b1; ++b) last = std::max(last, h[4 * b + 3]), st += h[4 * b + 3] ? (double)(h[4 * b + 2] - h[4 * b + 1]) : 0.0;
      if (last) fprintf(stderr, "[spam]   part %d: last block ends at %.1f us, stream time summed %.0f us\n", p, (double)(last - lo) / 100.0, st / 100.0);
    }
  }
  spmv_rg_reduce_kernel<T><<<(g->ntiles + 3) / 4, 256, 0, s>>>((const RgTile*)g->tiles, g->ntiles, g->mask_slot, g->slot_rows,
                                                              g->grp_cp, (T*)g->cp, y);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T>
int rg_refresh_values(const spam_csr* a, const spam_csr* res, RgPlan* g, hipStream_t s) {
  if (g->padded == 0) return SPAM_OK;
  if (!g->vsrc) return set_error(SPAM_E_INVALID, "row-group plan without value-source map (the caller rebuilds it)");
  rg_refresh_kernel<T><<<(unsigned)((g->padded + 255) / 256), 256, 0, s>>>(
      g->pe, (const RgBlk*)g->blk, g->ng * SPMV_NPART + g->ngs, g->ng * SPMV_NSLAB, g->ng * SPMV_NPART, g->vsrc,
      (const T*)a->vals, g->ngs ? (const T*)res->vals : nullptr, g->stream, g->padded);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

#define SPAM_RG_INSTANTIATE(T)                                                                                              \
  template int rg_build<T>(const spam_csr*, const uint8_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t,            \
                           const spam_csr*, uint32_t, const RgKnobs&, hipStream_t, RgPlan**);                                \
  template int rg_run<T>(const RgPlan*, const T*, uint32_t, const T*, uint32_t*, T*, const uint32_t*, hipStream_t);          \
  template int rg_refresh_values<T>(const spam_csr*, const spam_csr*, RgPlan*, hipStream_t);
SPAM_RG_INSTANTIATE(double)
SPAM_RG_INSTANTIATE(float)
SPAM_RG_INSTANTIATE(i64w)
#undef SPAM_RG_INSTANTIATE

}  // namespace spam
