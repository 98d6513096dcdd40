// spam_spgemm.hip — two-phase hash SpGEMM C = A B for gfx950.
//
// Reference counterpart: CsrMatrix::mul_hash (spam_csr/src/mul_hash.rs:13-201) with its linprobe
// accumulator (linprobe/src/{lib,set,map}.rs):
//   rows_to_threads  :38-64   -> spgemm_flops_kernel; partition rule in part_points_kernel (spam_rows_to_parts*,
//                                spam_spgemm_rank_step)
//   mul_hash_symbolic:66-103  -> sym_hash_kernel (rows of <= 512 products, several rows per workgroup),
//                                sym_row_kernel (<= 4096, one row per workgroup), sym_heavy_kernel (bitmap windows in
//                                LDS, the product streaming of the numeric heavy kernel)
//   checked_inclusive_scan (lib.rs:267-274) -> scan_u32_to_u64
//   mul_hash_numeric :106-201 -> num_hash_kernel / num_row_kernel (LDS linprobe map: u32 keys + T values, atomicCAS
//                                claim + LDS atomic add) for rows of up to 4096 entries; beyond: heavy_groups_kernel
//                                (work items) + num_heavy_kernel (rank-addressed LDS accumulator: bitmap + prefix popcount)
//   B2 (sort rows)   :164-175 -> sort_rows_kernel (bitonic in LDS); heavy rows are produced in column order
//   linprobe slot order (map.rs:59-63) -> ref_order_kernel / minpos_heavy_kernel + ref_order_long_kernel (sort_rows = 2)
//
// Where the CPU code gives each thread a contiguous flop-balanced row range (mul_hash.rs:52-62), the GPU bins rows
// by work: by F (intermediate products) for the symbolic pass, by nnz(C_i) for the numeric pass.  A bin fixes the
// lanes per row (16 / 64 / 256 / 1024) and the LDS table size; the hash bins keep linprobe's sentinel
// u32::MAX and its sizing max(16, 2 * next_pow2(n)) but hash with Fibonacci's constant (see home_slot).  Rows whose table would not fit LDS
// take the "heavy" path: the column range is cut into chunks, B's sorted rows are pre-split at the chunk boundaries
// (bpos) so a chunk range touches only its own products, a row's chunks are packed into groups that fit the LDS
// accumulator, and every (row, group) becomes a work item of a device-wide queue ordered by column window (see the
// comment above num_heavy_kernel).  No MFMA: nothing here is a dense contraction.
//
// Output order: sort_rows = 0 this library's own order (hash rows: slot order of a table filled in parallel; heavy
// rows: column order), 1 sorted by column (B2), 2 the reference's exact linprobe slot order.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "spam_internal.hpp"

namespace spam {

// ---- geometry -------------------------------------------------------------------------------------
// numeric chunk width of the heavy path = 16 columns per thread of the workgroup: 2^13 columns with 512 threads
// (72 KB of LDS: two workgroups per CU, one drains while the other multiplies — the default), 2^14 with 1024 threads
// (144 KB, one per CU: used for very wide B, where it halves the B-split table) and 2^12 with 256 (36 KB, four per CU).
// SPAM_HEAVY_TPB = 256 / 512 / 1024 forces one (A/B runs).
constexpr int CW_SHIFT_256 = 12, CW_SHIFT_512 = 13, CW_SHIFT_1024 = 14;
constexpr uint64_t WIDE_B_CHUNKS = 8192;        // more 2^13-column chunks than this: use the 2^14 geometry

// LDS accumulator of the heavy numeric path: 144 bytes per thread (two 512-thread workgroups per CU)
__host__ __device__ constexpr uint32_t heavy_region_bytes(int tpb) { return (uint32_t)tpb * 144u; }

constexpr uint32_t HEAVY_GRID = 8192;           // upper bound of the grids of the heavy-row helper kernels
// symbolic heavy kernel: 512 threads per (row, 2^18-column window) item, four workgroups per CU (8 waves per SIMD);
// 1024 threads / 2^19 columns for B wider than 4 M columns, 1024 / 2^20 (one per CU) beyond 16 M.  SPAM_SYM_GEOM =
// 0..3 forces one (tests, A/B runs).

constexpr int NSYM = 4;                         // symbolic bins by F
constexpr uint64_t SYM_MAX[NSYM - 1] = {64, 512, 4096};
constexpr int NNUM = 5;                         // numeric bins by nnz(C_i)
constexpr uint64_t NUM_MAX[NNUM - 1] = {64, 512, 2048, 4096};

struct BinLimits {
  uint64_t max[7];
  int nbins;  // bins 0..nbins-2 have max[], bin nbins-1 takes the rest; key 0 -> no bin
};

__device__ __forceinline__ uint32_t next_pow2_u32(uint32_t x) { return x <= 1 ? 1u : 1u << (32 - __clz(x - 1)); }
// linprobe sizing: max(16, 2 * next_power_of_two(n))  (set.rs:38-43, map.rs:33-38)
__device__ __forceinline__ uint32_t table_size_for(uint32_t n) {
  uint32_t t = 2u * next_pow2_u32(n);
  return t < MIN_TABLE ? MIN_TABLE : t;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d, 64));
  return v;
}
// the same sum through DPP moves (no LDS crossbar round trips); the total is returned in every lane
__device__ __forceinline__ uint32_t wave_sum_dpp_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- L1: per-row intermediate-product count (mul_hash.rs:39-50) -------------------------------------
// One wavefront per row; B.row_ptr gathers are the traffic.  total[0] += F, total[1] |= overflow.
__global__ __launch_bounds__(256) void spgemm_flops_kernel(const uint64_t* __restrict__ a_rp,
                                                          const uint32_t* __restrict__ a_col,
                                                          const uint64_t* __restrict__ b_rp, uint32_t rows,
                                                          uint64_t* __restrict__ flops,
                                                          unsigned long long* __restrict__ total) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  uint64_t mine = 0;
  for (uint32_t r = wave; r < rows; r += nwaves) {
    const uint64_t s = a_rp[r], e = a_rp[r + 1];
    uint64_t f = 0;
    for (uint64_t p = s + lane; p < e; p += 64) {
      const uint32_t k = a_col[p];
      f += b_rp[k + 1] - b_rp[k];
    }
    f = wave_sum_u64(f);
    if (lane == 0) {
      flops[r] = f;
      mine += f;
    }
  }
  __shared__ unsigned long long bsum;
  if (threadIdx.x == 0) bsum = 0;
  __syncthreads();
  if (lane == 0 && mine) atomicAdd(&bsum, (unsigned long long)mine);
  __syncthreads();
  if (threadIdx.x == 0 && bsum) {
    unsigned long long old = atomicAdd(&total[0], bsum);
    if (old + bsum < old) atomicOr(&total[1], 1ull);  // checked_add overflow (mul_hash.rs:47-48, lib.rs:270)
  }
}

// ---- row binning ------------------------------------------------------------------------------------
// lists[b * rows + i] = i-th row of bin b (order inside a bin is unspecified and does not affect results).
template <class KeyT>
__global__ __launch_bounds__(256) void bin_rows_kernel(const KeyT* __restrict__ key, uint32_t rows, BinLimits lim,
                                                      uint32_t* __restrict__ counts, uint32_t* __restrict__ lists,
                                                      uint32_t* __restrict__ pos_in_last /* may be null */) {
  __shared__ uint32_t lcnt[8], lbase[8];
  if (threadIdx.x < 8) lcnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  int b = -1;
  uint32_t rank = 0;
  if (r < rows) {
    const uint64_t k = (uint64_t)key[r];
    if (k != 0) {
      b = lim.nbins - 1;
      for (int i = lim.nbins - 2; i >= 0; --i)
        if (k <= lim.max[i]) b = i;
      rank = atomicAdd(&lcnt[b], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < lim.nbins && lcnt[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&counts[threadIdx.x], lcnt[threadIdx.x]);
  __syncthreads();
  if (b >= 0) {
    const uint32_t pos = lbase[b] + rank;
    lists[(uint64_t)b * rows + pos] = r;
    if (pos_in_last && b == lim.nbins - 1) pos_in_last[r] = pos;
  }
}

// LDS float add.  ds_add_f64 runs at ~2.7 lane-ops/clk/CU, but ds_add_f32 only at 0.38 on gfx950 while a CAS loop on
// the bit pattern reaches 2.4 (tools/lds_atomic_bench.hip, profiles/r01_lds_atomic_microbench.txt) — so f32 uses the loop.
__device__ __forceinline__ void lds_add(double* p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void lds_add(i64w* p, i64w v) { atomicAdd(p, v); }  // ds_add_u64: wrapping, order-independent
__device__ __forceinline__ void lds_add(float* p, float v) {
  uint32_t* q = (uint32_t*)p;
  uint32_t old = *q, assumed;
  do {
    assumed = old;
    old = atomicCAS(q, assumed, __float_as_uint(__uint_as_float(assumed) + v));
  } while (old != assumed);
}

// ---- linprobe in LDS --------------------------------------------------------------------------------
// Home slot of a key.  FIB = false: linprobe's (key * 107) & mask (lib.rs:13-14) — it maps columns c and c + size to
// the same slot, and R-MAT rows hold many such pairs, so linear probing degenerates into long runs.  FIB = true:
// Fibonacci hashing on the top bits; the tables of the hash bins use it (scale-20 A*A: the two one-row-per-workgroup
// numeric kernels 5.4 + 3.1 -> 3.0 + 1.7 ms).  Which slot a key lands in is visible only as the order of an UNSORTED
// output row (sort_rows = 0, compared as a set, tests.rs:356-371); the reference's exact slot order is sort_rows = 2,
// computed by ref_order_kernel with linprobe's own hash.
template <bool FIB>
__device__ __forceinline__ uint32_t home_slot(uint32_t key, uint32_t mask) {
  return FIB ? ((key * 0x9E3779B1u) >> (__clz(mask))) & mask : (key * HASH_SCAL) & mask;
}
constexpr bool BIN_FIB = true;

// linprobe::HashSet::insert (set.rs:128-160) with the claim made atomic: returns true when the key is new.
__device__ __forceinline__ bool lds_set_insert(uint32_t* table, uint32_t mask, uint32_t key) {
  uint32_t h = home_slot<BIN_FIB>(key, mask);
  for (;;) {
    const uint32_t old = atomicCAS(&table[h], EMPTY_KEY, key);
    if (old == EMPTY_KEY) return true;
    if (old == key) return false;
    h = (h + 1) & mask;
  }
}
// linprobe::HashMap::entry + and_modify/or_insert (map.rs:67-120; mul_hash.rs:157-161).  Values start at 0,
// so "first product stored" and "+= product" are the same atomic add.
template <class T, bool FIB = BIN_FIB>
__device__ __forceinline__ void lds_map_upsert(uint32_t* keys, T* vals, uint32_t mask, uint32_t key, T v) {
  uint32_t h = home_slot<FIB>(key, mask);
  for (;;) {
    const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, key);
    if (old == EMPTY_KEY || old == key) {
      lds_add(&vals[h], v);
      return;
    }
    h = (h + 1) & mask;
  }
}

template <int U>
__device__ __forceinline__ uint32_t lds_set_insert_batch(uint32_t* table, uint32_t mask, const uint32_t (&k)[U]) {
  uint32_t h[U], old[U], added = 0;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    h[j] = home_slot<BIN_FIB>(k[j], mask);
    old[j] = k[j] != EMPTY_KEY ? atomicCAS(&table[h[j]], EMPTY_KEY, k[j]) : k[j];
  }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    if (k[j] != EMPTY_KEY) {
      if (old[j] == EMPTY_KEY) {
        ++added;
      } else if (old[j] != k[j]) {
        uint32_t hh = (h[j] + 1) & mask;
        for (;;) {
          const uint32_t o = atomicCAS(&table[hh], EMPTY_KEY, k[j]);
          if (o == EMPTY_KEY) {
            ++added;
            break;
          }
          if (o == k[j]) break;
          hh = (hh + 1) & mask;
        }
      }
    }
  }
  return added;
}

// First probe of U independent upserts issued back to back (one LDS round trip for all of them); only the
// keys whose home slot holds another key fall back to the sequential probe loop.
template <class T, int U, bool FIB>
__device__ __forceinline__ void lds_map_upsert_batch(uint32_t* keys, T* vals, uint32_t mask, const uint32_t (&k)[U],
                                                     const T (&v)[U], const bool (&ok)[U]) {
  uint32_t h[U], old[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    h[j] = home_slot<FIB>(k[j], mask);
    old[j] = ok[j] ? atomicCAS(&keys[h[j]], EMPTY_KEY, k[j]) : k[j];
  }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    if (ok[j]) {
      if (old[j] == EMPTY_KEY || old[j] == k[j]) {
        lds_add(&vals[h[j]], v[j]);
      } else {
        uint32_t hh = (h[j] + 1) & mask;
        for (;;) {
          const uint32_t o = atomicCAS(&keys[hh], EMPTY_KEY, k[j]);
          if (o == EMPTY_KEY || o == k[j]) {
            lds_add(&vals[hh], v[j]);
            break;
          }
          hh = (hh + 1) & mask;
        }
      }
    }
  }
}

// ---- L4: symbolic, hash bins (mul_hash.rs:66-103) ----------------------------------------------------
// G lanes cooperate on one row; sub-groups of S lanes each take one A entry k and stride over B's row k.
template <int G, int S, int TABLE>
__global__ __launch_bounds__((G > 256 ? G : 256)) void sym_hash_kernel(
    const uint64_t* __restrict__ a_rp, const uint32_t* __restrict__ a_col, const uint64_t* __restrict__ b_rp,
    const uint32_t* __restrict__ b_col, const uint64_t* __restrict__ flops, const uint32_t* __restrict__ rows_list,
    uint32_t nrows, uint32_t* __restrict__ row_nnz) {
  constexpr int TPB = (G > 256 ? G : 256);
  constexpr int RPW = TPB / G;
  __shared__ uint32_t table[RPW][TABLE];
  __shared__ uint32_t cnt[RPW];
  const int g = threadIdx.x / G, gl = threadIdx.x % G;
  const uint32_t ridx = blockIdx.x * RPW + g;
  const bool valid = ridx < nrows;
  const uint32_t row = valid ? rows_list[ridx] : 0;
  uint32_t tsize = 0;
  if (valid) {
    const uint64_t f = flops[row];
    tsize = table_size_for((uint32_t)(f > TABLE / 2 ? TABLE / 2 : f));  // hs.shrink_to(row_nz) (mul_hash.rs:87)
    for (uint32_t i = gl; i < tsize; i += G) table[g][i] = EMPTY_KEY;
  }
  if (gl == 0) cnt[g] = 0;
  __syncthreads();
  if (valid) {
    const uint32_t mask = tsize - 1;
    const int sub = gl / S, sl = gl % S;
    uint32_t mine = 0;
    for (uint64_t p = a_rp[row] + sub, pe = a_rp[row + 1]; p < pe; p += G / S) {
      const uint32_t k = a_col[p];
      for (uint64_t q = b_rp[k] + sl, qe = b_rp[k + 1]; q < qe; q += S) mine += lds_set_insert(table[g], mask, b_col[q]);
    }
    if (mine) atomicAdd(&cnt[g], mine);
  }
  __syncthreads();
  if (valid && gl == 0) row_nnz[row] = cnt[g];  // *row_nz = hs.len() (mul_hash.rs:95)
}

// ---- flattened product space of the one-row-per-workgroup kernels -------------------------------------
// A batch of up to SEG entries k of A's row gives SEG segments [lo_s, lo_s + len_s) of B's arrays.  Their
// lengths are prefix-summed into LDS (seg_start) and the workgroup then walks the concatenated product
// index space: every thread takes several products per round (one binary search, then a forward walk over
// segment boundaries), so lanes stay busy whatever the mix of short and long B rows, and each lane has that
// many independent loads in flight.

// exclusive prefix + block total in every thread (two barriers, none needed afterwards to read the total)
template <int TPB>
__device__ __forceinline__ uint32_t block_excl_scan_tot(uint32_t v, uint32_t* wsum /* [TPB/64 + 1] */, uint32_t& total) {
  constexpr int NW = TPB / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();  // wsum may still be read from the previous call
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t c = wsum[i];
    if (i < w) off += c;
    tot += c;
  }
  total = tot;
  return off + inc - v;
}

template <int TPB>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wsum /* [TPB/64 + 1] */) {
  constexpr int NW = TPB / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();  // wsum may still be read from the previous call
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t c = wsum[i];
    if (i < w) off += c;
    tot += c;
  }
  if (threadIdx.x == 0) wsum[NW] = tot;
  return off + inc - v;
}


// largest s in [0, n) with seg_start[s] <= idx   (seg_start is non-decreasing, seg_start[0] = 0)
__device__ __forceinline__ uint32_t seg_find(const uint32_t* seg_start, uint32_t n, uint32_t idx) {
  uint32_t lo = 0, hi = n;  // invariant: seg_start[lo] <= idx, answer in [lo, hi)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (seg_start[mid] <= idx)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// forward variant: the answer is >= sg; walk a few segments, then bisect the rest
__device__ __forceinline__ uint32_t seg_advance(const uint32_t* seg_start, uint32_t n, uint32_t sg, uint32_t idx) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (idx < seg_start[sg + 1]) return sg;
    ++sg;
  }
  uint32_t lo = sg, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (seg_start[mid] <= idx)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// ---- one row per workgroup, flattened product space (mid-size rows) -----------------------------------
// Same traversal as the heavy kernels (segments = whole B rows, prefix sum in LDS, each wavefront takes
// 64*ROW_U consecutive products per round so every load is lane-contiguous), linprobe table in LDS.
constexpr int ROW_U = 8;

template <int TPB, int TABLE>
__global__ __launch_bounds__(TPB) void sym_row_kernel(const uint64_t* __restrict__ a_rp,
                                                     const uint32_t* __restrict__ a_col,
                                                     const uint64_t* __restrict__ b_rp,
                                                     const uint32_t* __restrict__ b_col,
                                                     const uint64_t* __restrict__ flops,
                                                     const uint32_t* __restrict__ rows_list, uint32_t nrows,
                                                     uint32_t* __restrict__ row_nnz) {
  constexpr int SEG = TPB;
  __shared__ uint32_t table[TABLE];
  __shared__ uint32_t seg_start[SEG + 1];
  __shared__ uint32_t seg_lo[SEG];
  __shared__ uint32_t wsum[TPB / 64 + 1];
  __shared__ uint32_t cnt;
  for (uint32_t ri = blockIdx.x; ri < nrows; ri += gridDim.x) {
    const uint32_t row = rows_list[ri];
    const uint64_t f = flops[row];
    const uint32_t tsize = table_size_for((uint32_t)(f > TABLE / 2 ? TABLE / 2 : f));  // hs.shrink_to (mul_hash.rs:87)
    const uint32_t mask = tsize - 1;
    for (uint32_t i = threadIdx.x; i < tsize; i += TPB) table[i] = EMPTY_KEY;
    if (threadIdx.x == 0) cnt = 0;
    uint32_t mine = 0;
    const uint64_t ps = a_rp[row], pe = a_rp[row + 1];
    for (uint64_t pb = ps; pb < pe; pb += SEG) {
      const uint64_t p = pb + threadIdx.x;
      uint32_t lo = 0, len = 0;
      if (p < pe) {
        const uint32_t k = a_col[p];
        lo = (uint32_t)b_rp[k];
        len = (uint32_t)b_rp[k + 1] - lo;
      }
      uint32_t total;
      const uint32_t st = block_excl_scan_tot<TPB>(len, wsum, total);
      seg_start[threadIdx.x] = st;
      seg_lo[threadIdx.x] = lo - st;  // q = seg_lo[s] + idx for a product index idx of segment s
      if (threadIdx.x == 0) seg_start[SEG] = total;
      __syncthreads();
      const uint32_t upl = total >= TPB * ROW_U ? ROW_U : (total + TPB - 1) / TPB;  // spread small rows over all waves
      for (uint32_t wb = (threadIdx.x >> 6) * (64 * upl); wb < total; wb += TPB * upl) {
        const uint32_t idx0 = wb + (threadIdx.x & 63);
        uint32_t sg = seg_find(seg_start, SEG, idx0 < total ? idx0 : total - 1);
        uint32_t cols[ROW_U];
#pragma unroll
        for (int j = 0; j < ROW_U; ++j) {
          const uint32_t idx = idx0 + j * 64;
          cols[j] = EMPTY_KEY;
          if (j < upl && idx < total) {
            sg = seg_advance(seg_start, SEG, sg, idx);
            cols[j] = b_col[seg_lo[sg] + idx];
          }
        }
        mine += lds_set_insert_batch<ROW_U>(table, mask, cols);
      }
      __syncthreads();
    }
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) row_nnz[row] = cnt;  // *row_nz = hs.len() (mul_hash.rs:95)
    __syncthreads();
  }
}

template <class T, int TPB, int TABLE>
__global__ __launch_bounds__(TPB) void num_row_kernel(
    const uint64_t* __restrict__ a_rp, const uint32_t* __restrict__ a_col, const T* __restrict__ a_val,
    const uint64_t* __restrict__ b_rp, const uint32_t* __restrict__ b_col, const T* __restrict__ b_val,
    const uint32_t* __restrict__ rows_list, uint32_t nrows, const uint32_t* __restrict__ row_nnz,
    const uint64_t* __restrict__ c_rp, uint32_t* __restrict__ c_col, T* __restrict__ c_val) {
  constexpr int SEG = TPB;
  __shared__ uint32_t keys[TABLE];
  __shared__ T vals[TABLE];
  __shared__ uint32_t seg_start[SEG + 1];
  __shared__ uint32_t seg_lo[SEG];
  __shared__ T seg_av[SEG];
  __shared__ uint32_t wsum[TPB / 64 + 1];
  for (uint32_t ri = blockIdx.x; ri < nrows; ri += gridDim.x) {
    const uint32_t row = rows_list[ri];
    const uint32_t tsize = table_size_for(row_nnz[row]);  // hm.shrink_to(*row_nz) (mul_hash.rs:144)
    const uint32_t mask = tsize - 1;
    for (uint32_t i = threadIdx.x; i < tsize; i += TPB) {
      keys[i] = EMPTY_KEY;
      vals[i] = acc_identity<T>();  // the first add then equals "store the first product" (mul_hash.rs:157-161), -0.0 included
    }
    const uint64_t ps = a_rp[row], pe = a_rp[row + 1];
    for (uint64_t pb = ps; pb < pe; pb += SEG) {
      const uint64_t p = pb + threadIdx.x;
      uint32_t lo = 0, len = 0;
      if (p < pe) {
        const uint32_t k = a_col[p];
        seg_av[threadIdx.x] = a_val[p];
        lo = (uint32_t)b_rp[k];
        len = (uint32_t)b_rp[k + 1] - lo;
      }
      uint32_t total;
      const uint32_t st = block_excl_scan_tot<TPB>(len, wsum, total);
      seg_start[threadIdx.x] = st;
      seg_lo[threadIdx.x] = lo - st;  // q = seg_lo[s] + idx for a product index idx of segment s
      if (threadIdx.x == 0) seg_start[SEG] = total;
      __syncthreads();
      const uint32_t upl = total >= TPB * ROW_U ? ROW_U : (total + TPB - 1) / TPB;  // spread small rows over all waves
      for (uint32_t wb = (threadIdx.x >> 6) * (64 * upl); wb < total; wb += TPB * upl) {
        const uint32_t idx0 = wb + (threadIdx.x & 63);
        uint32_t sg = seg_find(seg_start, SEG, idx0 < total ? idx0 : total - 1);
        uint32_t cols[ROW_U];
        T prods[ROW_U];
#pragma unroll
        for (int j = 0; j < ROW_U; ++j) {
          const uint32_t idx = idx0 + j * 64;
          cols[j] = EMPTY_KEY;
          prods[j] = T(0);
          if (j < upl && idx < total) {
            sg = seg_advance(seg_start, SEG, sg, idx);
            const uint32_t q = seg_lo[sg] + idx;
            cols[j] = b_col[q];
            prods[j] = seg_av[sg] * b_val[q];  // t * t1 (mul_hash.rs:154)
          }
        }
        bool okv[ROW_U];
#pragma unroll
        for (int j = 0; j < ROW_U; ++j) okv[j] = cols[j] != EMPTY_KEY;
        lds_map_upsert_batch<T, ROW_U, BIN_FIB>(keys, vals, mask, cols, prods, okv);
      }
      __syncthreads();
    }
    // drain in slot order (map.rs:59-63): thread t owns slots [t*spt, (t+1)*spt)
    const uint32_t spt = tsize / TPB;  // >= 1: this bin has tsize >= TPB
    uint32_t n = 0;
    for (uint32_t j = 0; j < spt; ++j) n += keys[threadIdx.x * spt + j] != EMPTY_KEY;
    const uint32_t ex = block_excl_scan<TPB>(n, wsum);
    uint64_t out = c_rp[row] + ex;
    for (uint32_t j = 0; j < spt; ++j) {
      const uint32_t slot = threadIdx.x * spt + j;
      const uint32_t k = keys[slot];
      if (k != EMPTY_KEY) {
        c_col[out] = k;
        c_val[out] = vals[slot];
        ++out;
      }
    }
    __syncthreads();
  }
}

// ---- heavy rows: B rows pre-split at numeric-chunk boundaries ---------------------------------------
// bpos[j * rows_b + k] = first position q in B's row k with col[q] >= j * CW, j = 0..nchunks (absolute, u32).
// Two kernels share the table: rows of up to BPOS_SHORT entries are walked by one lane each, chunk by chunk, so that the
// 64 rows of a wavefront write 64 consecutive table entries per chunk (the table is chunk-major: written row by row
// every entry is a 64-byte line of its own — 63 ms for the 8.6 GB table of a scale-22 B); longer rows (a few percent
// of them, most of the entries) take one wavefront each, which finds the chunk boundaries among its lanes' entries.
constexpr uint32_t BPOS_SHORT = 64;

// (round 4) Two changes.  (1) The walk needs each row's next chunk number at every step; read from HBM that was one
// dependent load per step and lane — 1.3 ms for a table whose 0.54 GB of stores take 0.15.  The chunk numbers of the 64
// rows of a wavefront are therefore staged in LDS first (64 independent gathers per lane at most, stride 65 words per
// lane: conflict-free when the cursors move together) and the walk reads them from there.  (2) The rows of more than
// BPOS_SHORT entries used to fill their column of the table themselves, one 4-byte store per chunk into a line the
// short rows' wavefront had left a hole in (1.35 ms: a read-modify-write of a line per store).  Now bpos_long_kernel
// writes the boundaries of every long row into a compact row-major side array (one wavefront per row, coalesced) and
// the walk kernel's lanes that hold a long row read their boundaries from there, so every store of the table is a
// full 256-byte line of 64 consecutive rows.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v);
constexpr uint32_t BPOS_SEG = 1024;  // entries of a long row one wavefront looks at (a hub row of 50 000 entries walked by ONE
                                     // wavefront was the critical path of the whole table: 780 dependent round trips, 1.25 ms)
struct BposItem {
  uint32_t k, seg;
};
// One thread per row: long rows get a slot in the side array and one work item per BPOS_SEG entries.
__global__ __launch_bounds__(256) void bpos_long_items_kernel(const uint64_t* __restrict__ b_rp, uint32_t rows_b,
                                                             uint32_t* __restrict__ counters /* [0] long rows, [1] items */,
                                                             uint32_t* __restrict__ lslot, BposItem* __restrict__ items) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint64_t len = 0;
  if (k < rows_b) len = b_rp[k + 1] - b_rp[k];
  const bool is_long = len > BPOS_SHORT;
  const uint32_t nseg = is_long ? (uint32_t)((len + BPOS_SEG - 1) / BPOS_SEG) : 0u;
  // one atomic per wavefront and counter (40 000 long rows adding to two addresses one by one took 0.13 ms)
  const uint64_t bal = __ballot(is_long);
  if (!bal) return;
  const uint32_t inc = wave_incl_scan_u32(nseg), tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  uint32_t slot0 = 0, base0 = 0;
  if (lane == 0) slot0 = atomicAdd(&counters[0], (uint32_t)__popcll(bal)), base0 = atomicAdd(&counters[1], tot);
  slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot0), base0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)base0);
  if (!is_long) return;
  lslot[k] = slot0 + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
  const uint32_t base = base0 + inc - nseg;
  for (uint32_t i = 0; i < nseg; ++i) items[base + i] = BposItem{k, i};
}
// One wavefront per item: the chunk boundaries that fall into its segment of the row, written to the row's line of lb.
__global__ __launch_bounds__(256) void bpos_long_kernel(const uint64_t* __restrict__ b_rp, const uint32_t* __restrict__ b_col,
                                                       uint32_t nchunks, uint32_t lb_stride, int CW_SHIFT,
                                                       const uint32_t* __restrict__ counters, const uint32_t* __restrict__ lslot,
                                                       const BposItem* __restrict__ items, uint32_t* __restrict__ lb) {
  const int lane = threadIdx.x & 63;
  const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= counters[1]) return;
  const BposItem it = items[w];
  const uint64_t s = b_rp[it.k], e = b_rp[it.k + 1];
  const uint64_t q0 = s + (uint64_t)it.seg * BPOS_SEG, q1 = q0 + BPOS_SEG < e ? q0 + BPOS_SEG : e;
  uint32_t* out = lb + (uint64_t)lslot[it.k] * lb_stride;
  for (uint64_t q = q0 + lane; q < q1; q += 64) {
    const uint32_t cq = b_col[q] >> CW_SHIFT;
    const uint32_t cp = (q == s) ? 0u : (b_col[q - 1] >> CW_SHIFT);
    for (uint32_t j = cp + 1; j <= cq; ++j) out[j] = (uint32_t)q;
  }
  if (it.seg == 0 && lane == 0) out[0] = (uint32_t)s;
  if (q1 == e) {  // (uniform) the chunks behind the row's last entry
    const uint32_t cl = b_col[e - 1] >> CW_SHIFT;
    for (uint32_t j = cl + 1 + lane; j <= nchunks; j += 64) out[j] = (uint32_t)e;
  }
}

__global__ __launch_bounds__(64) void bpos_walk_kernel(const uint64_t* __restrict__ b_rp, const uint32_t* __restrict__ b_col,
                                                      uint32_t rows_b, uint32_t nchunks, uint32_t lb_stride, int CW_SHIFT,
                                                      const uint32_t* __restrict__ lslot, const uint32_t* __restrict__ lb,
                                                      uint32_t* __restrict__ bpos) {
  __shared__ uint32_t st[64 * (BPOS_SHORT + 1)];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = 0, e = 0;
  if (k < rows_b) s = b_rp[k], e = b_rp[k + 1];
  const bool in = k < rows_b, is_long = in && e - s > BPOS_SHORT;
  const uint32_t len = in && !is_long ? (uint32_t)(e - s) : 0u;
  // a long row's boundaries come from its line of lb, four at a time (16 bytes per lane), read one group ahead
  const u32x4* mine_lb = reinterpret_cast<const u32x4*>(lb + (is_long ? (uint64_t)lslot[k] * lb_stride : 0ull));
  uint32_t* my = st + threadIdx.x * (BPOS_SHORT + 1);
  const uint32_t maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(len));
  for (uint32_t i0 = 0; i0 < maxlen; i0 += 8) {
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = i0 + u < len ? b_col[s + i0 + u] >> CW_SHIFT : 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < len) my[i0 + u] = c[u];
  }
  // (each lane reads back only what it wrote itself: no barrier)
  uint32_t cur = 0;
  uint32_t nc = cur < len ? my[cur] : 0xFFFFFFFFu;  // chunk of the entry at the cursor
  u32x4 nxt = {0u, 0u, 0u, 0u};
  if (is_long) nxt = mine_lb[0];
  for (uint32_t j0 = 0; j0 <= nchunks; j0 += 4) {
    const u32x4 lbv = nxt;
    if (is_long && j0 + 4 <= nchunks) nxt = mine_lb[(j0 >> 2) + 1];  // (lb lines are padded to a multiple of 4 words)
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t j = j0 + u;
      if (j > nchunks) break;
      while (nc < j) {
        ++cur;
        nc = cur < len ? my[cur] : 0xFFFFFFFFu;
      }
      uint32_t v = (uint32_t)(s + cur);
      if (is_long) v = lbv[u];
      if (in) bpos[(uint64_t)j * rows_b + k] = v;
    }
  }
}

// per heavy row: hoff = exclusive prefix of hcnt over chunks, row_nnz = total.  One wavefront per row.
__global__ __launch_bounds__(256) void heavy_finalize_kernel(const uint32_t* __restrict__ heavy_rows, uint32_t n_heavy,
                                                            uint32_t nchunks, const uint32_t* __restrict__ hcnt,
                                                            uint32_t* __restrict__ hoff, uint32_t* __restrict__ row_nnz) {
  const int lane = threadIdx.x & 63;
  const uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (h >= n_heavy) return;
  uint32_t run = 0;
  for (uint32_t base = 0; base < nchunks; base += 64) {
    const uint32_t c = base + lane;
    const uint32_t v = c < nchunks ? hcnt[(uint64_t)h * nchunks + c] : 0;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (c < nchunks) hoff[(uint64_t)h * nchunks + c] = run + inc - v;
    run += __shfl(inc, 63, 64);
  }
  if (lane == 0) row_nnz[heavy_rows[h]] = run;
}

// ---- L5: numeric, hash bins (mul_hash.rs:106-201) ----------------------------------------------------
template <class T, int G, int S, int TABLE>
__global__ __launch_bounds__((G > 256 ? G : 256)) void num_hash_kernel(
    const uint64_t* __restrict__ a_rp, const uint32_t* __restrict__ a_col, const T* __restrict__ a_val,
    const uint64_t* __restrict__ b_rp, const uint32_t* __restrict__ b_col, const T* __restrict__ b_val,
    const uint32_t* __restrict__ rows_list, uint32_t nrows, const uint32_t* __restrict__ row_nnz,
    const uint64_t* __restrict__ c_rp, uint32_t* __restrict__ c_col, T* __restrict__ c_val) {
  constexpr int TPB = (G > 256 ? G : 256);
  constexpr int RPW = TPB / G;
  __shared__ uint32_t keys[RPW][TABLE];
  __shared__ T vals[RPW][TABLE];
  __shared__ uint32_t wcnt[TPB / 64];
  const int g = threadIdx.x / G, gl = threadIdx.x % G;
  const uint32_t ridx = blockIdx.x * RPW + g;
  const bool valid = ridx < nrows;
  const uint32_t row = valid ? rows_list[ridx] : 0;
  uint32_t tsize = 0;
  if (valid) {
    tsize = table_size_for(row_nnz[row]);  // hm.shrink_to(*row_nz) (mul_hash.rs:144)
    for (uint32_t i = gl; i < tsize; i += G) {
      keys[g][i] = EMPTY_KEY;
      vals[g][i] = acc_identity<T>();  // (-0.0: the first add equals "store the first product", mul_hash.rs:157-161)
    }
  }
  __syncthreads();
  if (valid) {
    const uint32_t mask = tsize - 1;
    const int sub = gl / S, sl = gl % S;
    for (uint64_t p = a_rp[row] + sub, pe = a_rp[row + 1]; p < pe; p += G / S) {
      const uint32_t k = a_col[p];
      const T av = a_val[p];
      for (uint64_t q = b_rp[k] + sl, qe = b_rp[k + 1]; q < qe; q += S)
        lds_map_upsert<T>(keys[g], vals[g], mask, b_col[q], av * b_val[q]);  // t * t1 (mul_hash.rs:154)
    }
  }
  __syncthreads();
  // drain in slot order (map.rs:59-63): position = number of occupied slots before this one
  const uint64_t out0 = valid ? c_rp[row] : 0;
  if (G <= 64) {
    if (valid) {
      const int lane = threadIdx.x & 63;
      const int gshift = lane & ~(G - 1);
      const uint64_t gmask = (G == 64) ? ~0ull : ((1ull << (G & 63)) - 1);
      uint32_t run = 0;
      for (uint32_t base = 0; base < tsize; base += G) {
        const uint32_t slot = base + gl;
        const uint32_t key = keys[g][slot];
        const bool occ = key != EMPTY_KEY;
        const uint64_t m = (__ballot(occ) >> gshift) & gmask;
        if (occ) {
          const uint32_t pos = run + __popcll(m & ((1ull << gl) - 1));
          c_col[out0 + pos] = key;
          c_val[out0 + pos] = vals[g][slot];
        }
        run += __popcll(m);
      }
    }
  } else {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t run = 0;
    for (uint32_t base = 0; base < tsize; base += G) {  // one row per workgroup: uniform trip count
      const uint32_t slot = base + gl;
      const uint32_t key = slot < tsize ? keys[g][slot] : EMPTY_KEY;
      const bool occ = key != EMPTY_KEY;
      const uint64_t m = __ballot(occ);
      if (lane == 0) wcnt[w] = __popcll(m);
      __syncthreads();
      uint32_t off = 0, tot = 0;
#pragma unroll
      for (int i = 0; i < TPB / 64; ++i) {
        const uint32_t c = wcnt[i];
        if (i < w) off += c;
        tot += c;
      }
      if (occ) {
        const uint32_t pos = run + off + __popcll(m & ((1ull << lane) - 1));
        c_col[out0 + pos] = key;
        c_val[out0 + pos] = vals[g][slot];
      }
      run += tot;
      __syncthreads();
    }
  }
}

// Longest-first order for the work queues: rows are bucketed by floor(log2(F)) (64 buckets, heaviest bucket
// first; order inside a bucket is unspecified).  A queue that ends on small rows has a short tail.
__global__ __launch_bounds__(256) void cost_hist_kernel(const uint32_t* __restrict__ list, uint32_t n,
                                                       const uint64_t* __restrict__ cost, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&h[63 - (63 - __clzll(cost[list[i]] | 1ull))], 1u);  // bucket 0 = largest costs
  __syncthreads();
  if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void cost_scan_kernel(uint32_t* __restrict__ hist /* [64] -> exclusive prefix */) {
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int b = 0; b < 64; ++b) {
      const uint32_t c = hist[b];
      hist[b] = run;
      run += c;
    }
  }
}
__global__ __launch_bounds__(256) void cost_scatter_kernel(const uint32_t* __restrict__ list, uint32_t n,
                                                          const uint64_t* __restrict__ cost, uint32_t* __restrict__ cursor,
                                                          uint32_t* __restrict__ out) {
  __shared__ uint32_t h[64], base[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t r = 0, b = 0, rank = 0;
  if (i < n) {
    r = list[i];
    b = 63 - (63 - __clzll(cost[r] | 1ull));
    rank = atomicAdd(&h[b], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64 && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  if (i < n) out[base[b] + rank] = r;
}

// ---- heavy numeric path: wave-autonomous product streaming into a rank-addressed accumulator ------------------
// A heavy row (more than 4096 entries of C) is cut into GROUPS of consecutive column chunks; every (row, group) is a
// work item of its own.  A group's entries are accumulated in LDS in two passes over its products:
//   pass 1  marks the columns that occur:  ds_or on a bitmap over the group's column span (fire and forget);
//   then every bitmap word gets the number of set bits before it (its entries' first output position);
//   pass 2  adds each product to vals[rank(col)], rank = word prefix + popcount of the lower bits: one ds_read_b64 and one
//           ds_add_f64, no compare-and-swap, no probing — the linprobe table this replaces spent most of its time in
//           data-dependent CAS loops (its longest probe sequence among 64 lanes sets the pace of the wavefront);
//   drain   vals[0 .. cnt) is already the row piece in COLUMN order: one coalesced copy, the column indices are expanded
//           from the bitmap.  Heavy rows therefore come out sorted whatever sort_rows says.
// LDS holds the bitmap words {prefix, bits} followed by the values: 8 * span / 32 + sizeof(T) * cnt bytes must fit
// REGION, which bounds a group either by its entries (dense stretches: ~8000 entries) or by its span (sparse tails);
// heavy_groups_kernel packs chunks greedily.  Items are queued in COLUMN-WINDOW order (all groups starting in chunk 0,
// then chunk 1, ...), so at any moment the workgroups of the whole chip read the same narrow slice of B's rows, which
// stays in every XCD's L2, instead of all of B.  Workgroups take items in runs of ITEM_RUN consecutive queue
// positions (one atomic per run) and prefetch across items: while item i is multiplied, the B-split lookups of item
// i + 1 and the A entries of item i + 2 are in flight.
// Inside an item the wavefronts stream products on their own (no workgroup barrier inside a pass): the first 64
// entries of A's row (in a sorted row the lowest columns: R-MAT's hubs, whose B rows hold most of the products) are
// SHARED — every wavefront holds all of them and takes its share of each segment; the entries after them are OWNED,
// thread (wave w, lane l) owns entry 64 + l * NW + w; rows longer than 64 + TPB entries stream the rest in 64-entry
// batches per wavefront.  Segments of 64 or more products are cut into 64-product PIECES with scalar parameters (one
// coalesced load per piece, no search); shorter ones are walked through a flattened product index space.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// inclusive prefix sum over the 64 lanes (Hillis-Steele inside the 16-lane rows, then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += dpp_u32<0x111, 0xF>(v);  // row_shr:1
  v += dpp_u32<0x112, 0xF>(v);  // row_shr:2
  v += dpp_u32<0x114, 0xF>(v);  // row_shr:4
  v += dpp_u32<0x118, 0xF>(v);  // row_shr:8
  v += dpp_u32<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
  v += dpp_u32<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, uint32_t src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
__device__ __forceinline__ double shfl_val(double v, uint32_t src_lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const uint32_t lo = shfl_u32((uint32_t)b, src_lane), hi = shfl_u32((uint32_t)(b >> 32), src_lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
__device__ __forceinline__ float shfl_val(float v, uint32_t src_lane) {
  return __builtin_bit_cast(float, shfl_u32(__builtin_bit_cast(uint32_t, v), src_lane));
}
__device__ __forceinline__ i64w shfl_val(i64w v, uint32_t src_lane) {
  const uint32_t lo = shfl_u32((uint32_t)v, src_lane), hi = shfl_u32((uint32_t)(v >> 32), src_lane);
  return ((i64w)hi << 32) | (i64w)lo;
}
__device__ __forceinline__ double readlane_val(double v, uint32_t src_lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, (int)src_lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), (int)src_lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
__device__ __forceinline__ i64w readlane_val(i64w v, uint32_t src_lane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src_lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)src_lane);
  return ((i64w)hi << 32) | (i64w)lo;
}
__device__ __forceinline__ float readlane_val(float v, uint32_t src_lane) {
  return __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_readlane((int)__builtin_bit_cast(uint32_t, v), (int)src_lane));
}

#ifndef SPAM_WAVE_U
#define SPAM_WAVE_U 4
#endif
constexpr int WAVE_U = SPAM_WAVE_U;  // products per lane and round: that many independent (col, val) loads in flight per lane
#ifndef SPAM_LONG_SEG
#define SPAM_LONG_SEG 64
#endif
constexpr uint32_t LONG_SEG = SPAM_LONG_SEG;  // segments of at least this many products are cut into 64-product pieces
#ifndef SPAM_PIECE_U
#define SPAM_PIECE_U 8
#endif
constexpr int PIECE_U = SPAM_PIECE_U;  // pieces per round of the long-segment path in pass 2 (column + value loads)
#ifndef SPAM_PIECE_U1
#define SPAM_PIECE_U1 8
#endif
constexpr int PIECE_U1 = SPAM_PIECE_U1;  // ... and in pass 1 (column loads only)
#ifndef SPAM_FLAT_EARLY
#define SPAM_FLAT_EARLY 1
#endif
constexpr bool FLAT_EARLY = SPAM_FLAT_EARLY != 0;
#ifndef SPAM_TICKET
#define SPAM_TICKET 1
#endif
constexpr bool TICKETS = SPAM_TICKET != 0;
#ifndef SPAM_TICKET_FROM_PASS
#define SPAM_TICKET_FROM_PASS 2
#endif
constexpr int TICKET_FROM_PASS = SPAM_TICKET_FROM_PASS;  // numeric heavy kernel: piece rounds of the shared head handed out by LDS ticket  // pass 1: first short-segment round in flight during the piece rounds

struct RankWord {  // one bitmap word of the group's column span
  uint32_t prefix;  // entries of the group in lower words (valid after the prefix step)
  uint32_t bits;
};

// One round of WAVE_U products per lane.  PASS 0 (symbolic) / 1: mark the columns; PASS 2: add the products at their rank.
// A lane without a product passes the column c0 - 1 (no_col): relative to the window that is 0xFFFFFFFF, outside like
// any column the B-split did not filter, so one compare serves both.
__device__ __forceinline__ uint32_t no_col(uint32_t c0) { return c0 - 1u; }

template <class T, int PASS, int U>
__device__ __forceinline__ void accumulate_round(const uint32_t (&cols)[U], const T (&prods)[U], uint32_t c0,
                                                 uint32_t cspan, RankWord* words, T* vals) {
  if (PASS == 0) {
    // symbolic phase: `words` is a plain bitmap, one bit per column of the window — transposed inside every block of
    // 4096 columns (128 words): column r of the block is bit r / 128 of word r % 128.  Only the bit COUNT of a chunk (a
    // whole number of blocks) is read back, and the 64 consecutive columns of a dense B row's piece then go to 64
    // different words instead of two (same-address LDS atomics serialise: 59 % of the LDS busy time before).
    uint32_t* bm = reinterpret_cast<uint32_t*>(words);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t rel = cols[j] - c0;
      if (rel < cspan) atomicOr(&bm[((rel >> 12) << 7) | (rel & 127u)], 1u << ((rel >> 7) & 31u));
    }
  } else if (PASS == 1) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t rel = cols[j] - c0;
      if (rel < cspan) atomicOr(&words[rel >> 5].bits, 1u << (rel & 31));
    }
  } else {
    RankWord w[U];
    uint32_t slot[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t rel = cols[j] - c0;
      w[j] = words[rel < cspan ? rel >> 5 : 0u];  // unconditional (word 0 for the idle lanes): all reads in flight together
    }
    // the ranks first, unconditionally and in straight-line code, the adds afterwards: an add issued between two rank
    // computations makes the next one's wait for its word (lgkmcnt) wait for that add's round trip as well
#pragma unroll
    for (int j = 0; j < U; ++j) {
      slot[j] = w[j].prefix + __popc(w[j].bits & ((1u << ((cols[j] - c0) & 31)) - 1u));
      asm volatile("" : "+v"(slot[j]));  // (keeps the compiler from sinking the rank into the guarded add below)
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t rel = cols[j] - c0;
      if (rel < cspan) lds_add(&vals[slot[j]], prods[j]);
    }
  }
}

// Long segments: 64-product pieces.  `deal` = NW for the shared entries every wavefront holds (piece p of segment s goes
// to wavefront (s + p) mod NW, so a hub's products spread evenly over the workgroup) and 1 for entries this wavefront
// owns alone.  The wavefront's pieces are numbered through (lane s contributes cnt[s] of them, prefix-summed with DPP
// moves) and lane t builds the descriptor of piece base + t: its segment by the same lockstep bisection the short
// segments use — once per 64 pieces —, then position, length and A's value by ds_bpermute.  A round reads PIECE_U
// descriptors back as scalars (v_readlane at a uniform index): a piece is one coalesced load of columns (and one of
// values in pass 2) at a scalar base + lane, and all loads of the round are in flight together.  (A scalar iterator
// walking the segments piece by piece cost more issue slots than the products: ~60 scalar instructions and a dozen
// branches per piece.)
template <class T, int PASS>
__device__ __forceinline__ void long_pieces(uint32_t lo, uint32_t len, T av, uint32_t deal, uint32_t wave,
                                            const uint32_t* __restrict__ b_col, const T* __restrict__ b_val, uint32_t c0,
                                            uint32_t cspan, RankWord* words, T* vals) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t np = len >= LONG_SEG ? (len + 63) >> 6 : 0u;                       // pieces of this lane's segment
  const uint32_t p0 = deal > 1 ? (wave + deal - (lane % deal)) % deal : 0u;         // this wavefront's first one
  const uint32_t cnt = p0 < np ? (np - 1 - p0) / deal + 1 : 0u;                     // ... and how many: p0, p0 + deal, ...
  const uint32_t inc = wave_incl_scan_u32(cnt);
  const uint32_t st = inc - cnt;
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  constexpr int PU = PASS <= 1 ? PIECE_U1 : PIECE_U;
  for (uint32_t base = 0; base < total; base += 64) {
    const uint32_t nb = total - base < 64 ? total - base : 64;  // descriptors of this batch (uniform)
    const uint32_t g = base + lane;
    // the segment of piece g: the last lane whose start is <= g (st is non-decreasing; lanes without pieces share
    // their successor's start)
    const uint32_t w1 = base + nb - 1;
    uint32_t slo = (uint32_t)__popcll(__ballot(st <= base)) - 1u, shi = (uint32_t)__popcll(__ballot(st <= w1)) - 1u;
    for (uint32_t span = shi - slo; span != 0; span >>= 1) {  // uniform trip count
      const uint32_t mid = (slo + shi + 1) >> 1;
      if (shfl_u32(st, mid) <= g)
        slo = mid;
      else
        shi = mid - 1;
    }
    const uint32_t k = g - shfl_u32(st, slo);
    const uint32_t pp = (shfl_u32(p0, slo) + k * deal) * 64;  // first product of the piece inside its segment
    uint32_t q0 = shfl_u32(lo, slo) + pp;
    uint32_t n = shfl_u32(len, slo) - pp;
    T a = PASS == 2 ? shfl_val(av, slo) : T(0);
    if (lane >= nb) n = 0;
    for (uint32_t r = 0; r < nb; r += PU) {
      uint32_t cols[PU];
      T prods[PU], sa[PU];
#pragma unroll
      for (int j = 0; j < PU; ++j) {
        const uint32_t dl = (r + j) & 63;  // (lanes >= nb hold n = 0; the index only wraps when 64 is no multiple of PU)
        const uint32_t qj = (uint32_t)__builtin_amdgcn_readlane((int)q0, (int)dl);
        const uint32_t nj = (64 % PU == 0 || r + j < 64) ? (uint32_t)__builtin_amdgcn_readlane((int)n, (int)dl) : 0u;
        sa[j] = PASS == 2 ? readlane_val(a, dl) : T(0);
        cols[j] = no_col(c0);
        prods[j] = T(0);
        if (lane < nj) {
          const uint64_t q = (uint64_t)qj + lane;
          cols[j] = b_col[q];
          if (PASS == 2) prods[j] = b_val[q];
        }
      }
      if (PASS == 2) {  // only now: a multiply right behind its load would make every piece wait for its own round trip
#pragma unroll
        for (int j = 0; j < PU; ++j) prods[j] = sa[j] * prods[j];  // t * t1 (mul_hash.rs:154)
      }
      accumulate_round<T, PASS, PU>(cols, prods, c0, cspan, words, vals);
    }
  }
}

// The same for the SHARED entries with the rounds handed out dynamically: all wavefronts number the pieces of the long
// segments through identically, and a wavefront that is ready for more work takes the next round of PU consecutive
// pieces with an LDS ticket (the passes of an item end when its slowest wavefront does, and with a fixed share per
// wavefront that one takes 1.24x the mean: the spread comes from the memory round trips, not from the piece counts).
// The PU pieces of a round lie in a handful of neighbouring segments, so their search is zero to two halvings.
template <class T, int PASS>
__device__ __forceinline__ void ticket_pieces(uint32_t lo, uint32_t len, T av, uint32_t* ticket,
                                              const uint32_t* __restrict__ b_col, const T* __restrict__ b_val, uint32_t c0,
                                              uint32_t cspan, RankWord* words, T* vals) {
  const uint32_t lane = threadIdx.x & 63;
  constexpr int PU = PASS <= 1 ? PIECE_U1 : PIECE_U;
  const uint32_t np = len >= LONG_SEG ? (len + 63) >> 6 : 0u;  // pieces of this lane's segment
  const uint32_t inc = wave_incl_scan_u32(np);
  const uint32_t st = inc - np;
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  const uint32_t nrounds = (total + PU - 1) / PU;
  auto grab = [&]() -> uint32_t {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
  };
  // descriptors of round r in the lanes 0 .. PU - 1: position, length and A's value of piece r * PU + lane
  auto build = [&](uint32_t r, uint32_t& q0, uint32_t& n, T& a) {
    const uint32_t g0 = r * PU, g1 = (g0 + PU - 1 < total ? g0 + PU - 1 : total - 1);
    const uint32_t g = g0 + lane < total ? g0 + lane : total - 1;  // (lanes >= PU compute along, unused)
    uint32_t slo = (uint32_t)__popcll(__ballot(st <= g0)) - 1u, shi = (uint32_t)__popcll(__ballot(st <= g1)) - 1u;
    for (uint32_t span = shi - slo; span != 0; span >>= 1) {  // uniform trip count
      const uint32_t mid = (slo + shi + 1) >> 1;
      if (shfl_u32(st, mid) <= g)
        slo = mid;
      else
        shi = mid - 1;
    }
    const uint32_t pp = (g - shfl_u32(st, slo)) * 64;  // first product of the piece inside its segment
    q0 = shfl_u32(lo, slo) + pp;
    n = shfl_u32(len, slo) - pp;
    a = PASS == 2 ? shfl_val(av, slo) : T(0);
    if (g0 + lane >= total) n = 0;
  };
  uint32_t r = grab();
  uint32_t q0 = 0, n = 0;
  T a = T(0);
  if (r < nrounds) build(r, q0, n, a);
  while (r < nrounds) {
    uint32_t cols[PU];
    T prods[PU], sa[PU];
#pragma unroll
    for (int j = 0; j < PU; ++j) {
      const uint32_t qj = (uint32_t)__builtin_amdgcn_readlane((int)q0, j);
      const uint32_t nj = (uint32_t)__builtin_amdgcn_readlane((int)n, j);
      sa[j] = PASS == 2 ? readlane_val(a, (uint32_t)j) : T(0);
      cols[j] = no_col(c0);
      prods[j] = T(0);
      if (lane < nj) {
        const uint64_t q = (uint64_t)qj + lane;
        cols[j] = b_col[q];
        if (PASS == 2) prods[j] = b_val[q];
      }
    }
    // the next round's ticket and descriptors under this round's loads
    r = grab();
    if (r < nrounds) build(r, q0, n, a);
    if (PASS == 2) {
#pragma unroll
      for (int j = 0; j < PU; ++j) prods[j] = sa[j] * prods[j];  // t * t1 (mul_hash.rs:154)
    }
    accumulate_round<T, PASS, PU>(cols, prods, c0, cspan, words, vals);
  }
}

// The products of one wavefront's 64 segments (one per lane: positions [lo, lo + len) of B's arrays, value av).
// shared = true: every wavefront of the workgroup holds the same 64 segments (the head of A's row): long ones are
// dealt piecewise, a short one goes whole to wavefront lane mod NW.  shared = false: the segments are this wavefront's
// own.  Short segments are walked through the flattened product index space: the lengths are prefix-summed across the
// wave with DPP moves and the wave takes 64 * WAVE_U consecutive products per round, lane-contiguous inside every load.
// The segment of a product index is found by a search over the lanes' start offsets: two ballots bound the segments a
// 64-product window touches (a scalar range), the bisection inside that range runs for all windows of the round at
// once, one ds_bpermute round trip per halving — none when a window lies in one segment.
template <class T, int PASS>
__device__ __forceinline__ void wave_products(uint32_t lo, uint32_t len, T av, bool shared, uint32_t nw, uint32_t wave,
                                              const uint32_t* __restrict__ b_col, const T* __restrict__ b_val, uint32_t c0,
                                              uint32_t cspan, RankWord* words, T* vals, uint32_t* ticket = nullptr) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t lmask = __ballot(len >= LONG_SEG);
  // this lane's short segment.  Of the shared entries a wavefront takes the short segments of the lanes wave, wave + nw,
  // ...: they are moved to the lanes 0 .. 64 / nw - 1 first, so that the segment search below bisects over the 64 / nw
  // candidates instead of over lane indices nw apart (three halvings less per round)
  uint32_t slen = len >= LONG_SEG ? 0u : len, flo = lo;
  T fav = av;
  if (shared) {
    const uint32_t src = (lane * nw + wave) & 63;
    const uint32_t l2 = shfl_u32(slen, src);
    flo = shfl_u32(lo, src);
    if (PASS == 2) fav = shfl_val(av, src);
    slen = lane < 64 / nw ? l2 : 0u;
  }
  const uint32_t inc = wave_incl_scan_u32(slen);
  const uint32_t st = inc - slen;                        // first product index of this lane's segment
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  const uint32_t off = flo - st;                         // q = off[s] + idx for a product index idx of segment s
  // one round of the short segments, in two halves: the segment search and the loads ...
  auto issue = [&](uint32_t base, uint32_t (&slo)[WAVE_U], uint32_t (&cols)[WAVE_U], T (&bv)[WAVE_U]) {
    // st is non-decreasing over the lanes: the lanes with st <= w form a prefix whose last member is the segment
    // holding product w (empty segments share their successor's start and are skipped by taking the last)
    uint32_t shi[WAVE_U];
    uint32_t maxspan = 0;
#pragma unroll
    for (int j = 0; j < WAVE_U; ++j) {
      const uint32_t w0 = base + 64 * j;                 // uniform
      slo[j] = 0, shi[j] = 0;
      if (w0 < total) {
        const uint32_t w1 = (w0 + 63 < total ? w0 + 63 : total - 1);
        const uint32_t a0 = (uint32_t)__popcll(__ballot(st <= w0)) - 1u;
        const uint32_t a1 = (uint32_t)__popcll(__ballot(st <= w1)) - 1u;
        slo[j] = a0, shi[j] = a1;
        maxspan = a1 - a0 > maxspan ? a1 - a0 : maxspan;
      }
    }
#ifdef SPAM_ABLATE_SEARCH  // (timing experiments only: results are wrong)
    maxspan = 0;
#endif
    for (uint32_t span = maxspan; span != 0; span >>= 1) {  // uniform trip count
      uint32_t v[WAVE_U], mid[WAVE_U];
#pragma unroll
      for (int j = 0; j < WAVE_U; ++j) {
        mid[j] = (slo[j] + shi[j] + 1) >> 1;
        v[j] = shfl_u32(st, mid[j]);
      }
#pragma unroll
      for (int j = 0; j < WAVE_U; ++j) {
        const uint32_t idx = base + 64 * j + lane;
        if (v[j] <= idx)
          slo[j] = mid[j];                               // (mid == slo when the range is a single segment: no change)
        else
          shi[j] = mid[j] - 1;
      }
    }
    uint32_t qq[WAVE_U];
#pragma unroll
    for (int j = 0; j < WAVE_U; ++j) qq[j] = shfl_u32(off, slo[j]);
#pragma unroll
    for (int j = 0; j < WAVE_U; ++j) {
      const uint32_t idx = base + 64 * j + lane;
      cols[j] = no_col(c0);
      bv[j] = T(0);
      if (idx < total) {
        cols[j] = b_col[qq[j] + idx];
        if (PASS == 2) bv[j] = b_val[qq[j] + idx];
      }
    }
  };
  // ... and the accumulation
  auto finish = [&](const uint32_t (&slo)[WAVE_U], const uint32_t (&cols)[WAVE_U], const T (&bv)[WAVE_U]) {
    T aa[WAVE_U], prods[WAVE_U];
#pragma unroll
    for (int j = 0; j < WAVE_U; ++j) aa[j] = PASS == 2 ? shfl_val(fav, slo[j]) : T(0);
#pragma unroll
    for (int j = 0; j < WAVE_U; ++j) prods[j] = aa[j] * bv[j];  // t * t1 (mul_hash.rs:154)
    accumulate_round<T, PASS, WAVE_U>(cols, prods, c0, cspan, words, vals);
  };
  // The first round of the short segments is issued BEFORE the pieces of the long ones and accumulated after them: its
  // round trip to memory overlaps theirs (a wavefront has ~3 piece rounds and ~1.7 short rounds per pass, every one a
  // full memory latency).  Pass 1 only: in pass 2 the values would have to be held as well, and there are no
  // registers left for them.
  constexpr bool EARLY = FLAT_EARLY && PASS <= 1;
  uint32_t slo0[WAVE_U], cols0[WAVE_U];
  T bv0[WAVE_U];
  if (ticket) {  // (uniform) the wavefront's fixed share first, the ticketed rounds last: they even out what it took
    if (EARLY && total) issue(0, slo0, cols0, bv0);
    for (uint32_t base = EARLY ? 64 * WAVE_U : 0; base < total; base += 64 * WAVE_U) {
      uint32_t slo[WAVE_U], cols[WAVE_U];
      T bv[WAVE_U];
      issue(base, slo, cols, bv);
      finish(slo, cols, bv);
    }
    if (lmask) ticket_pieces<T, PASS>(lo, len, av, ticket, b_col, b_val, c0, cspan, words, vals);
    if (EARLY && total) finish(slo0, cols0, bv0);
    return;
  }
  if (EARLY && total) issue(0, slo0, cols0, bv0);
  if (lmask) long_pieces<T, PASS>(lo, len, av, shared ? nw : 1u, wave, b_col, b_val, c0, cspan, words, vals);
  if (EARLY && total) finish(slo0, cols0, bv0);
  for (uint32_t base = EARLY ? 64 * WAVE_U : 0; base < total; base += 64 * WAVE_U) {
    uint32_t slo[WAVE_U], cols[WAVE_U];
    T bv[WAVE_U];
    issue(base, slo, cols, bv);
    finish(slo, cols, bv);
  }
}

constexpr uint32_t ITEM_RUN = 8;  // queue positions a workgroup claims per atomic

// [lo, lo + len): the entries of B's row k with columns in the chunks [j0, j1) (the whole row without the pre-split)
template <bool USE_BPOS>
__device__ __forceinline__ void b_range(const uint32_t* __restrict__ bpos, const uint64_t* __restrict__ b_rp, uint32_t rows_b,
                                        uint32_t j0, uint32_t j1, uint32_t k, uint32_t& lo, uint32_t& len) {
  lo = 0, len = 0;
  if (k != 0xFFFFFFFFu) {
    if (USE_BPOS) {
      lo = bpos[(uint64_t)j0 * rows_b + k];
      len = bpos[(uint64_t)j1 * rows_b + k] - lo;
    } else {
      lo = (uint32_t)b_rp[k];
      len = (uint32_t)b_rp[k + 1] - lo;
    }
  }
}

// Rows of A longer than 64 + 64 * NW entries: after the shared head, wavefront w takes the 64-entry batches w, w + NW,
// ...; while batch i is multiplied the B-split lookups of batch i + 1 and the entries of batch i + 2 are in flight.
template <class T, int PASS, bool USE_BPOS, int NW>
__device__ __forceinline__ void streamed_products(const uint32_t* __restrict__ a_col, const T* __restrict__ a_val,
                                                  uint64_t a_start, uint32_t n_a, uint32_t j0, uint32_t j1, uint32_t wave,
                                                  const uint64_t* __restrict__ b_rp, const uint32_t* __restrict__ b_col,
                                                  const T* __restrict__ b_val, const uint32_t* __restrict__ bpos,
                                                  uint32_t rows_b, uint32_t c0, uint32_t cspan, RankWord* words, T* vals) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t ps = a_start + 64, pe = a_start + n_a;
  const uint64_t nb = ((uint64_t)n_a - 64 + 63) >> 6;
  auto s_entry = [&](uint64_t bb, uint32_t& k, T& a) {
    const uint64_t p = ps + (bb << 6) + lane;
    k = 0xFFFFFFFFu, a = T(0);
    if (bb < nb && p < pe) {
      k = a_col[p];
      if (PASS == 2) a = a_val[p];
    }
  };
  uint32_t k1, k2, lo1, len1;
  T a1, a2;
  s_entry(wave, k1, a1);
  s_entry(wave + NW, k2, a2);
  b_range<USE_BPOS>(bpos, b_rp, rows_b, j0, j1, k1, lo1, len1);
  for (uint64_t b = wave; b < nb; b += NW) {
    const uint32_t lo_b = lo1, len_b = len1;
    const T a_b = a1;
    b_range<USE_BPOS>(bpos, b_rp, rows_b, j0, j1, k2, lo1, len1);
    a1 = a2;
    s_entry(b + 2 * NW, k2, a2);
    wave_products<T, PASS>(lo_b, len_b, a_b, false, NW, wave, b_col, b_val, c0, cspan, words, vals);
  }
}

// ---- symbolic phase of the heavy rows (mul_hash.rs:66-103 for rows with more than 4096 intermediate products) ----------
// One workgroup per (row, column window): the window's columns are a bitmap in LDS (32 words per thread: 2^20 columns
// with 1024 threads), the products are streamed exactly as in the numeric phase below — per-wavefront, shared head of
// 64 entries dealt piecewise, owned entries, streamed batches — and marked with LDS atomic ORs; then every wavefront
// counts the bits of its chunks (hcnt[heavy row][chunk], what the group builder packs by) and clears them.  Items come
// from a device-wide queue in runs of ITEM_RUN, heaviest rows first; descriptors are read three items ahead, A's
// entries two, the B-split lookups one, so a workgroup never waits for the chain queue -> row -> entries -> ranges.
struct SymItem {     // 16 bytes, read with scalar loads
  uint64_t a_start;  // first entry of A's row
  uint32_t n_a;      // entries of A's row
  uint32_t hrow;     // index of the row among the heavy rows
};

__global__ __launch_bounds__(256) void sym_items_kernel(const uint32_t* __restrict__ order, uint32_t n,
                                                       const uint64_t* __restrict__ a_rp,
                                                       const uint32_t* __restrict__ hrow_of_row, SymItem* __restrict__ items) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t row = order[i];
  SymItem it;
  it.a_start = a_rp[row], it.n_a = (uint32_t)(a_rp[row + 1] - a_rp[row]), it.hrow = hrow_of_row[row];
  items[i] = it;
}

template <bool USE_BPOS, int TPB, int WIN_SHIFT_, int WAVES_PER_SIMD>
__global__ __launch_bounds__(TPB, WAVES_PER_SIMD) void sym_heavy_kernel(const uint32_t* __restrict__ a_col, const uint64_t* __restrict__ b_rp,
                                                       const uint32_t* __restrict__ b_col, const uint32_t* __restrict__ bpos,
                                                       uint32_t rows_b, const SymItem* __restrict__ items, uint32_t n_rows,
                                                       uint32_t nwin, uint32_t n_giant, uint32_t sub_chunks, uint32_t nchunks,
                                                       int cw_shift, uint32_t* __restrict__ hcnt, uint32_t* __restrict__ queue) {
  constexpr int NW = TPB / 64;
  constexpr uint32_t WIN_SHIFT = WIN_SHIFT_;
  constexpr uint32_t BM_WORDS = (1u << WIN_SHIFT) / 32;         // 128 KB for 2^20 columns
  __shared__ __attribute__((aligned(16))) uint32_t bm[BM_WORDS];
  __shared__ uint32_t run_s;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t ch_per_win = (1u << WIN_SHIFT) >> cw_shift;    // numeric chunks per window
  // the first n_giant rows of the list (the heaviest: a (row, window) item of theirs would keep one workgroup busy for
  // milliseconds while the rest of the chip has run dry) are cut into sub-windows of sub_chunks chunks instead
  // (round 4 measured sub-windows that double inside every window — 24 items per giant row instead of 128 — against
  // these uniform ones on the 8 rank blocks of scale 20: symbolic phase 4.1 against 4.0 ms per block, no difference)
  const uint32_t nsub = (nchunks + sub_chunks - 1) / sub_chunks;
  const uint32_t n_giant_items = n_giant * nsub;
  const uint32_t n_items = n_giant_items + (n_rows - n_giant) * nwin;  // (the host checks that this fits 32 bits)
  const uint32_t cwords = (1u << cw_shift) >> 5;                // bitmap words per chunk
  for (int i = threadIdx.x; i < (int)(BM_WORDS / 4); i += TPB) ((u32x4*)bm)[i] = (u32x4)(0u);
  if (threadIdx.x == 0) run_s = atomicAdd(queue, ITEM_RUN);
  __syncthreads();
  uint32_t run_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_s), run_next = 0xFFFFFFFFu, pos = 0;
  __syncthreads();
  struct Cur {
    uint64_t a_start;
    uint32_t n_a, hrow, j0, j1, live;  // [j0, j1): the item's chunks
  };
  auto item_index = [&](uint32_t ahead) -> uint32_t {  // queue position of the item `ahead` steps after the current one
    const uint32_t q = pos + ahead;
    if (q < ITEM_RUN) return run_base < n_items ? run_base + q : 0xFFFFFFFFu;
    return run_next < n_items ? run_next + (q - ITEM_RUN) : 0xFFFFFFFFu;  // (ahead <= 3: the next run is known from pos 1 on)
  };
  auto load_item = [&](uint32_t idx, Cur& it) {
    it.a_start = 0, it.n_a = 0, it.hrow = 0, it.j0 = 0, it.j1 = 0, it.live = 0;
    if (idx < n_items) {
      uint32_t r, j0, width;
      if (idx < n_giant_items) {
        r = idx / nsub;
        j0 = (idx - r * nsub) * sub_chunks, width = sub_chunks;
      } else {
        const uint32_t i2 = idx - n_giant_items, r2 = nwin == 1 ? i2 : i2 / nwin;
        r = n_giant + r2;
        j0 = (i2 - r2 * nwin) * ch_per_win, width = ch_per_win;
      }
      const SymItem si = items[r];  // uniform address: scalar loads
      it.a_start = si.a_start, it.n_a = si.n_a, it.hrow = si.hrow, it.live = 1;
      it.j0 = j0, it.j1 = j0 + width < nchunks ? j0 + width : nchunks;
    }
  };
  auto fetch_entry = [&](const Cur& it, uint32_t& ks, uint32_t& ko) {
    ks = 0xFFFFFFFFu, ko = 0xFFFFFFFFu;
    if (lane < it.n_a) ks = a_col[it.a_start + lane];
    const uint32_t e = 64 + lane * NW + wave;
    if (it.n_a <= (uint32_t)(64 + TPB) && e < it.n_a) ko = a_col[it.a_start + e];
  };
  auto fetch_range = [&](const Cur& it, uint32_t k, uint32_t& lo, uint32_t& len) {
    b_range<USE_BPOS>(bpos, b_rp, rows_b, it.j0, it.j1, k, lo, len);
  };
  Cur it_c, it_n, it_nn, it_n3;
  uint32_t ks_c, ks_n, ks_nn, los_c, lens_c, los_n, lens_n;  // shared entry: current / next / next-next item
  uint32_t ko_c, ko_n, ko_nn, loo_c, leno_c, loo_n, leno_n;  // owned entry
  load_item(item_index(0), it_c);
  load_item(item_index(1), it_n);
  load_item(item_index(2), it_nn);
  fetch_entry(it_c, ks_c, ko_c);
  fetch_entry(it_n, ks_n, ko_n);
  fetch_range(it_c, ks_c, los_c, lens_c);
  fetch_range(it_c, ko_c, loo_c, leno_c);
  while (it_c.live) {
    unsigned int claimed = 0;
    if (pos == 0 && threadIdx.x == 0) claimed = atomicAdd(queue, ITEM_RUN);
    const uint32_t j0 = it_c.j0, j1 = it_c.j1;
    const uint32_t c0 = j0 << cw_shift, cspan = (j1 - j0) << cw_shift;
    fetch_range(it_n, ks_n, los_n, lens_n);
    fetch_range(it_n, ko_n, loo_n, leno_n);
    load_item(item_index(3), it_n3);
    fetch_entry(it_nn, ks_nn, ko_nn);
    RankWord* const words = reinterpret_cast<RankWord*>(bm);  // (PASS 0 addresses it as the plain bitmap it is)
    wave_products<float, 0>(los_c, lens_c, 0.f, true, NW, wave, b_col, nullptr, c0, cspan, words, nullptr);
    if (it_c.n_a <= (uint32_t)(64 + TPB)) {
      if (it_c.n_a > 64)  // (uniform)
        wave_products<float, 0>(loo_c, leno_c, 0.f, false, NW, wave, b_col, nullptr, c0, cspan, words, nullptr);
    } else {
      streamed_products<float, 0, USE_BPOS, NW>(a_col, nullptr, it_c.a_start, it_c.n_a, j0, j1, wave, b_rp, b_col, nullptr, bpos,
                                               rows_b, c0, cspan, words, nullptr);
    }
    __syncthreads();  // the window's bitmap is complete
    // nnz per numeric chunk; chunks are dealt round-robin (the dense ones are neighbours); the bitmap is cleared on the way
    for (uint32_t c = wave; c < j1 - j0; c += NW) {
      uint32_t n = 0;
      for (uint32_t i = lane; i < cwords; i += 64) {
        const uint32_t v = bm[c * cwords + i];
        if (v) bm[c * cwords + i] = 0;
        n += __popc(v);
      }
      n = wave_sum_dpp_u32(n);
      if (lane == 0) hcnt[(uint64_t)it_c.hrow * nchunks + j0 + c] = n;
    }
    if (pos == 0 && threadIdx.x == 0) run_s = claimed;
    __syncthreads();  // LDS is all zero again; the claimed run is visible
    if (pos == 0) run_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_s);
    if (++pos == ITEM_RUN) {
      pos = 0;
      run_base = run_next;
      run_next = 0xFFFFFFFFu;
    }
    it_c = it_n, it_n = it_nn, it_nn = it_n3;
    ks_c = ks_n, los_c = los_n, lens_c = lens_n;
    ko_c = ko_n, loo_c = loo_n, leno_c = leno_n;
    ks_n = ks_nn, ko_n = ko_nn;
  }
}

struct HeavyItem {       // 32 bytes, read with scalar loads
  uint64_t a_start;      // first entry of A's row
  uint64_t out;          // where the group's piece of C's row starts
  uint32_t n_a;          // entries of A's row
  uint32_t j0, j1;       // chunk range [j0, j1) of the group
  uint32_t cnt;          // entries of C the group produces (> 0)
};

// Pack one heavy row's chunks into groups, left to right, while bitmap + values fit the accumulator (8 bytes per 32
// columns of span + vsize bytes per entry <= region), and call emit(j0, j1, first entry offset, entries) once per group
// (uniform arguments; every lane calls).  One wavefront per row.  The walk is per GROUP, not per chunk: for the 64
// chunks of a block the byte needs b_j = word bytes + c_j * vsize and the counts are prefix-summed across the wave; one
// ballot finds the first non-empty chunk that no longer fits the open group, everything before it joins.
template <class Emit>
__device__ __forceinline__ void walk_row_groups(const uint32_t* __restrict__ hc, const uint32_t* __restrict__ ho,
                                                uint32_t nchunks, uint32_t word_bytes_per_chunk, uint32_t vsize,
                                                uint32_t region, Emit&& emit) {
  const uint32_t lane = threadIdx.x & 63;
  bool open = false;
  uint32_t g_start = 0, g_off = 0, g_cnt = 0, g_last = 0;  // the open group; g_last = its last non-empty chunk
  long long g_bytes = 0;  // bytes of the open group up to lane l of the current block = g_bytes + binc[l]
  for (uint32_t base = 0; base < nchunks; base += 64) {
    const uint32_t j = base + lane;
    const bool in = j < nchunks;
    const uint32_t c_l = in ? hc[j] : 0u, o_l = in ? ho[j] : 0u;
    const uint32_t b_l = in ? word_bytes_per_chunk + c_l * vsize : 0u;
    const uint32_t binc = wave_incl_scan_u32(b_l), cinc = wave_incl_scan_u32(c_l);
    const uint64_t nz = __ballot(c_l != 0);
    uint32_t pos = 0;
    while (pos < 64) {
      const uint64_t m = nz & (~0ull << pos);
      if (!m) break;
      if (!open) {
        const uint32_t sl = (uint32_t)__builtin_ctzll(m);
        open = true;
        g_start = base + sl, g_cnt = 0, g_last = base + sl;
        g_off = (uint32_t)__builtin_amdgcn_readlane((int)o_l, (int)sl);
        g_bytes = -(long long)((uint32_t)__builtin_amdgcn_readlane((int)binc, (int)sl) -
                               (uint32_t)__builtin_amdgcn_readlane((int)b_l, (int)sl));
        pos = sl;
      }
      const uint32_t cex_pos = (uint32_t)__builtin_amdgcn_readlane((int)cinc, (int)pos) -
                               (uint32_t)__builtin_amdgcn_readlane((int)c_l, (int)pos);
      const uint64_t ov = __ballot(c_l != 0 && lane >= pos && g_bytes + (long long)binc > (long long)region);
      if (!ov) {  // every remaining non-empty chunk of the block joins
        g_cnt += (uint32_t)__builtin_amdgcn_readlane((int)cinc, 63) - cex_pos;
        g_last = base + 63u - (uint32_t)__builtin_clzll(m);
        break;
      }
      const uint32_t f = (uint32_t)__builtin_ctzll(ov);  // first non-empty chunk that does not fit: the group ends before it
      const uint64_t members = m & ((1ull << f) - 1ull);
      if (members) {
        g_cnt += ((uint32_t)__builtin_amdgcn_readlane((int)cinc, (int)f) - (uint32_t)__builtin_amdgcn_readlane((int)c_l, (int)f)) - cex_pos;
        g_last = base + 63u - (uint32_t)__builtin_clzll(members);
      }
      emit(g_start, g_last + 1, g_off, g_cnt);
      open = false;
      pos = f;
    }
    if (open) g_bytes += (long long)(uint32_t)__builtin_amdgcn_readlane((int)binc, 63);
  }
  if (open) emit(g_start, g_last + 1, g_off, g_cnt);
}

// Group the heavy rows and emit one item per group, bucketed by first chunk (= the queue order).  Nearly every row has
// a group starting in chunk 0, so per-item atomics on the bucket counters would serialise on a handful of addresses
// (measured: 8 ms for 2.3 M items); instead a workgroup walks its rows twice: first it counts its groups per bucket in
// LDS, reserves one range per non-empty bucket with ONE global atomic, then walks the rows again and places the items
// with LDS atomics.  FILL = false is the counting launch (hist[j0] += groups, census), FILL = true writes the items
// (cursor = exclusive prefix of hist).  HIST_CAP buckets fit LDS; wider matrices take the per-item global atomics.
constexpr uint32_t HIST_CAP = 4096;

template <bool FILL>
__global__ __launch_bounds__(256) void heavy_groups_kernel(const uint32_t* __restrict__ rows_list, uint32_t n_rows,
                                                          const uint32_t* __restrict__ hrow_of_row, uint32_t nchunks,
                                                          const uint32_t* __restrict__ hcnt,
                                                          const uint32_t* __restrict__ hoff,
                                                          const uint64_t* __restrict__ a_rp,
                                                          const uint64_t* __restrict__ c_rp, uint32_t cw_shift,
                                                          uint32_t vsize, uint32_t region,
                                                          uint32_t* __restrict__ hist_or_cursor,
                                                          HeavyItem* __restrict__ items,
                                                          unsigned long long* __restrict__ census) {
  extern __shared__ uint32_t lds_hist[];  // [2 * nchunks] when nchunks <= HIST_CAP: local counts | reserved bases
  const bool local = nchunks <= HIST_CAP;
  uint32_t* cnt_l = lds_hist;
  uint32_t* base_l = lds_hist + nchunks;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t word_bytes_per_chunk = ((1u << cw_shift) / 32u) * 8u;
  if (local) {
    for (uint32_t i = threadIdx.x; i < nchunks; i += blockDim.x) cnt_l[i] = 0;
    __syncthreads();
  }
  unsigned long long ng = 0, ne = 0;
  // pass 1: count this workgroup's groups per bucket
  for (uint32_t w = wave_g; w < n_rows; w += nwaves) {
    const uint32_t row = rows_list[w];
    const uint32_t* hc = hcnt + (uint64_t)hrow_of_row[row] * nchunks;
    const uint32_t* ho = hoff + (uint64_t)hrow_of_row[row] * nchunks;
    walk_row_groups(hc, ho, nchunks, word_bytes_per_chunk, vsize, region, [&](uint32_t j0, uint32_t, uint32_t, uint32_t cnt) {
      if (lane == 0) {
        if (local)
          atomicAdd(&cnt_l[j0], 1u);
        else if (!FILL)
          atomicAdd(&hist_or_cursor[j0], 1u);
      }
      ng += 1, ne += cnt;
    });
  }
  if (!FILL) {
    if (local) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < nchunks; i += blockDim.x)
        if (cnt_l[i]) atomicAdd(&hist_or_cursor[i], cnt_l[i]);
    }
    if (lane == 0 && ng) atomicAdd(&census[0], ng), atomicAdd(&census[2], ne);
    return;
  }
  if (local) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nchunks; i += blockDim.x) {
      const uint32_t c = cnt_l[i];
      base_l[i] = c ? atomicAdd(&hist_or_cursor[i], c) : 0u;
      cnt_l[i] = 0;
    }
    __syncthreads();
  }
  // pass 2: the same walk, now writing the items
  for (uint32_t w = wave_g; w < n_rows; w += nwaves) {
    const uint32_t row = rows_list[w];
    const uint32_t* hc = hcnt + (uint64_t)hrow_of_row[row] * nchunks;
    const uint32_t* ho = hoff + (uint64_t)hrow_of_row[row] * nchunks;
    const uint64_t a_start = a_rp[row], out_row = c_rp[row];
    const uint32_t n_a = (uint32_t)(a_rp[row + 1] - a_start);
    walk_row_groups(hc, ho, nchunks, word_bytes_per_chunk, vsize, region, [&](uint32_t j0, uint32_t j1, uint32_t off0, uint32_t cnt) {
      if (lane == 0) {
        const uint32_t pos = local ? base_l[j0] + atomicAdd(&cnt_l[j0], 1u) : atomicAdd(&hist_or_cursor[j0], 1u);
        HeavyItem it;
        it.a_start = a_start, it.out = out_row + off0, it.n_a = n_a, it.j0 = j0, it.j1 = j1, it.cnt = cnt;
        items[pos] = it;
      }
    });
  }
}

template <class T, bool USE_BPOS, int TPB, int CWS>
__global__ __launch_bounds__(TPB, 4) void num_heavy_kernel(  // 4 waves per SIMD: 16 wavefronts per CU whatever TPB is
    const uint32_t* __restrict__ a_col, const T* __restrict__ a_val, const uint64_t* __restrict__ b_rp,
    const uint32_t* __restrict__ b_col, const T* __restrict__ b_val, const uint32_t* __restrict__ bpos, uint32_t rows_b,
    const HeavyItem* __restrict__ items, uint32_t n_items, uint32_t* __restrict__ queue, uint32_t* __restrict__ c_col,
    T* __restrict__ c_val, unsigned long long* __restrict__ gstats, int dbg) {
  constexpr int NW = TPB / 64;
  constexpr int CW_SHIFT = CWS;
  constexpr uint32_t REGION = heavy_region_bytes(TPB);
  static_assert((1u << CWS) == 16u * TPB, "chunk width is 16 columns per thread");
  __shared__ __attribute__((aligned(16))) unsigned char region[REGION];
  __shared__ uint32_t wsum[NW + 1];
  __shared__ uint32_t run_s;
  __shared__ uint32_t tick[2];  // round tickets of the two passes (SPAM_TICKET)
  RankWord* words = (RankWord*)region;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  // (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is uniform over a wavefront, and everything derived
  // from it — piece dealing, word slices — would otherwise be emitted as divergent vector code)
  const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // phase timers of this workgroup (thread 0 only, SPAM_DBG_HEAVY=32), cycles: [0] prefetch issue, [1] pass 1 (wave 0),
  // [2] barrier after pass 1, [3] prefix step, [4] pass 2 (wave 0), [5] barrier after pass 2, [6] drain, [7] closing
  // barrier, [8] items
  __shared__ unsigned long long wg_stat[9];
  if (threadIdx.x < 9) wg_stat[threadIdx.x] = 0;
  unsigned long long t0 = 0;
  const bool timing = (dbg & 32) && threadIdx.x == 0;
  for (int i = threadIdx.x; i < (int)(REGION / 16); i += TPB) ((u32x4*)region)[i] = (u32x4)(0u);
  if (threadIdx.x == 0) run_s = atomicAdd(queue, ITEM_RUN), tick[0] = 0, tick[1] = 0;
  __syncthreads();
  // the workgroup's item sequence: runs of ITEM_RUN consecutive queue positions; `pos` walks the current run
  uint32_t run_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_s), run_next = 0xFFFFFFFFu, pos = 0;
  __syncthreads();
  auto item_index = [&](uint32_t ahead) -> uint32_t {  // queue position of the item `ahead` steps after the current one
    const uint32_t q = pos + ahead;
    if (q < ITEM_RUN) return run_base < n_items ? run_base + q : 0xFFFFFFFFu;
    return run_next < n_items ? run_next + (q - ITEM_RUN) : 0xFFFFFFFFu;  // (ahead <= 3: the next run is known from pos 1 on)
  };
  auto load_item = [&](uint32_t idx, HeavyItem& it) {
    it.n_a = 0, it.cnt = 0, it.j0 = 0, it.j1 = 0, it.a_start = 0, it.out = 0;
    if (idx < n_items) it = items[idx];  // uniform address: scalar loads
  };
  auto fetch_entry = [&](const HeavyItem& it, uint32_t& ks, T& as, uint32_t& ko, T& ao) {
    ks = 0xFFFFFFFFu, ko = 0xFFFFFFFFu, as = T(0), ao = T(0);
    if (lane < it.n_a) {
      ks = a_col[it.a_start + lane];
      as = a_val[it.a_start + lane];
    }
    const uint32_t e = 64 + lane * NW + wave;
    if (it.n_a <= (uint32_t)(64 + TPB) && e < it.n_a) {
      ko = a_col[it.a_start + e];
      ao = a_val[it.a_start + e];
    }
  };
  auto fetch_range = [&](uint32_t j0, uint32_t j1, uint32_t k, uint32_t& lo, uint32_t& len) {
    b_range<USE_BPOS>(bpos, b_rp, rows_b, j0, j1, k, lo, len);
  };
  HeavyItem it_c, it_n, it_nn, it_n3;  // descriptors are read two items before their entries are requested
  uint32_t ks_c, ks_n, ks_nn, los_c, lens_c, los_n, lens_n;  // shared entry: current / next / next-next item
  uint32_t ko_c, ko_n, ko_nn, loo_c, leno_c, loo_n, leno_n;  // owned entry
  T as_c, as_n, as_nn, ao_c, ao_n, ao_nn;
  load_item(item_index(0), it_c);
  load_item(item_index(1), it_n);
  load_item(item_index(2), it_nn);
  fetch_entry(it_c, ks_c, as_c, ko_c, ao_c);
  fetch_entry(it_n, ks_n, as_n, ko_n, ao_n);
  fetch_range(it_c.j0, it_c.j1, ks_c, los_c, lens_c);
  fetch_range(it_c.j0, it_c.j1, ko_c, loo_c, leno_c);
  while (it_c.cnt) {  // cnt == 0 marks the end of the queue
    if (timing) t0 = clock64();
    // the run after this one is claimed while the first item of a run is processed and read at its end
    unsigned int claimed = 0;
    if (pos == 0 && threadIdx.x == 0) claimed = atomicAdd(queue, ITEM_RUN);
    const uint32_t j0 = it_c.j0, j1 = it_c.j1;
    const uint32_t c0 = j0 << CW_SHIFT, cspan = (j1 - j0) << CW_SHIFT;
    const uint32_t nwords = cspan >> 5;                       // multiple of 16 * TPB / 32 = TPB / 2: a whole number of
    T* vals = (T*)(region + (size_t)nwords * sizeof(RankWord));  // 64-word steps for every wavefront
    // float accumulators start at -0.0, the additive identity that keeps the sign of a first product of -0.0 ("store the
    // first product", mul_hash.rs:157-161); pass 1 touches only the bitmap words, so these writes ride under its loads
    if constexpr (!std::is_integral<T>::value)
      for (uint32_t i = threadIdx.x; i < it_c.cnt; i += TPB) vals[i] = acc_identity<T>();
    // prefetch: B-split lookups of the next item (its entries were requested an item ago), entries of the one after
    fetch_range(it_n.j0, it_n.j1, ks_n, los_n, lens_n);
    fetch_range(it_n.j0, it_n.j1, ko_n, loo_n, leno_n);
    load_item(item_index(3), it_n3);
    fetch_entry(it_nn, ks_nn, as_nn, ko_nn, ao_nn);
    if (timing) {
      const unsigned long long t1 = clock64();
      wg_stat[0] += t1 - t0, wg_stat[8] += 1;
      t0 = t1;
    }
    // ---- both passes over the item's products: every wavefront on its own ------------------------------------
    auto all_products = [&](auto pass_tag) {
      constexpr int PASS = decltype(pass_tag)::value;
      constexpr bool TK = TICKETS && PASS >= TICKET_FROM_PASS;
      if (!TK) wave_products<T, PASS>(los_c, lens_c, as_c, true, NW, wave, b_col, b_val, c0, cspan, words, vals);
      if (it_c.n_a <= (uint32_t)(64 + TPB)) {
        if (it_c.n_a > 64)  // (uniform)
          wave_products<T, PASS>(loo_c, leno_c, ao_c, false, NW, wave, b_col, b_val, c0, cspan, words, vals);
      } else {
        streamed_products<T, PASS, USE_BPOS, NW>(a_col, a_val, it_c.a_start, it_c.n_a, j0, j1, wave, b_rp, b_col, b_val, bpos,
                                                 rows_b, c0, cspan, words, vals);
      }
      if (TK)  // the shared head last, its piece rounds by ticket
        wave_products<T, PASS>(los_c, lens_c, as_c, true, NW, wave, b_col, b_val, c0, cspan, words, vals, &tick[PASS - 1]);
    };
    all_products(std::integral_constant<int, 1>{});
    if (timing) {
      const unsigned long long t1 = clock64();
      wg_stat[1] += t1 - t0;
      t0 = t1;
    }
    __syncthreads();  // the bitmap is complete
    if (timing) {
      const unsigned long long t1 = clock64();
      wg_stat[2] += t1 - t0;
      t0 = t1;
    }
    // ---- prefix step: wavefront w owns the words [w * wpw, (w + 1) * wpw), 64 consecutive words per step ----------
    const uint32_t wpw = nwords / NW;  // nwords is a multiple of 64 * NW / 8 ... see below: handled with a tail check
    {
      uint32_t n = 0;
      for (uint32_t i = wave * wpw + lane; i < (wave + 1) * wpw; i += 64) n += __popc(words[i].bits);
      n = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(n), 63);
      if (lane == 0) wsum[wave] = n;
    }
    __syncthreads();
    {
      uint32_t run = 0;
#pragma unroll
      for (int i = 0; i < NW; ++i) run += (i < (int)wave) ? wsum[i] : 0u;
      for (uint32_t i0 = wave * wpw; i0 < (wave + 1) * wpw; i0 += 64) {  // (uniform trip count; wpw may not be a multiple of 64)
        const uint32_t i = i0 + lane;
        const bool in = i < (wave + 1) * wpw;
        const uint32_t pc = in ? __popc(words[i].bits) : 0u;
        const uint32_t inc = wave_incl_scan_u32(pc);
        if (in) words[i].prefix = run + inc - pc;
        run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      }
    }
    __syncthreads();  // every word knows its first output position
    if (timing) {
      const unsigned long long t1 = clock64();
      wg_stat[3] += t1 - t0;
      t0 = t1;
    }
    all_products(std::integral_constant<int, 2>{});
    if (timing) {
      const unsigned long long t1 = clock64();
      wg_stat[4] += t1 - t0;
      t0 = t1;
    }
    __syncthreads();  // all products of the group are in LDS
    if (timing) {
      const unsigned long long t1 = clock64();
      wg_stat[5] += t1 - t0;
      t0 = t1;
    }
    // ---- drain: the values are the row piece in column order; the columns are expanded from the bitmap -----------
    // vmcnt(0) here, where every load in flight (the prefetches for the coming items) has long landed: without it the
    // compiler, which cannot tell, puts that wait inside the store loops below, where it waits for the loop's own
    // stores to be acknowledged, once per trip.  (simm16: vmcnt 0, expcnt 7, lgkmcnt 15)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const uint64_t out = it_c.out;
    const uint32_t cnt = it_c.cnt;
    for (uint32_t i = threadIdx.x; i < cnt; i += TPB) {
      c_val[out + i] = vals[i];
      vals[i] = T(0);  // all-zero bits: the next item lays its bitmap words and values out differently over the same LDS
    }
    // columns: every lane expands its own words (a loop over the word's set bits; adjacent lanes hold adjacent words, so
    // the stores of one step land in one stretch of C).  Two alternatives measured slower: 32 lanes per word with the
    // word broadcast by ds_bpermute (30.8 k cycles per item against 11.7 k) or re-read from LDS (16.5 k).
    {
      const uint32_t wend = (wave + 1) * wpw;
      auto word_at = [&](uint32_t i) {
        RankWord w;
        w.prefix = 0, w.bits = 0;
        if (i < wend) w = words[i];
        return w;
      };
      RankWord w = word_at(wave * wpw + lane);
      for (uint32_t i0 = wave * wpw; i0 < wend; i0 += 64) {
        const uint32_t i = i0 + lane;
        const RankWord wn = word_at(i + 64);  // the next step's word: its LDS round trip overlaps this step's stores
        // (round 4: expanding DENSE steps — 256 / 512 / 1024 columns and more in the 64 words — one lane per column, with
        // v_readlane pairs as lane masks, v_mbcnt ranks and 32 coalesced stores per step, measured 1.3-2.9 ms SLOWER than
        // this per-word bit loop everywhere: 115.7 / 116.9 / 117.3 against 114.4 ms of numeric phase)
#ifdef SPAM_ABLATE_DRAINCOLS  // (timing experiments only: results are wrong)
        if (false) {
#else
        if (w.bits) {
#endif
          uint32_t bits = w.bits;
          const uint32_t cbase = c0 + (i << 5);
          uint64_t o = out + w.prefix;
          while (bits) {  // (round 4: two set bits per trip with 32-bit offsets from a uniform base measured 1.5 ms slower)
            c_col[o++] = cbase + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1;
          }
        }
        if (w.bits | w.prefix) {
          RankWord z;
          z.prefix = 0, z.bits = 0;
          words[i] = z;
        }
        w = wn;
      }
    }
    if (pos == 0 && threadIdx.x == 0) run_s = claimed;
    if (TICKETS && threadIdx.x == 0) tick[0] = 0, tick[1] = 0;  // (both passes are over; the next item starts behind the barrier)
    if (timing) {
      const unsigned long long t1 = clock64();
      wg_stat[6] += t1 - t0;
      t0 = t1;
    }
    __syncthreads();  // LDS is all zero again; the claimed run is visible
    if (pos == 0) run_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)run_s);
    if (timing) wg_stat[7] += clock64() - t0;
    // advance the sequence and shift the prefetch pipeline
    if (++pos == ITEM_RUN) {
      pos = 0;
      run_base = run_next;
      run_next = 0xFFFFFFFFu;
    }
    it_c = it_n, it_n = it_nn, it_nn = it_n3;
    ks_c = ks_n, as_c = as_n, los_c = los_n, lens_c = lens_n;
    ko_c = ko_n, ao_c = ao_n, loo_c = loo_n, leno_c = leno_n;
    ks_n = ks_nn, as_n = as_nn, ko_n = ko_nn, ao_n = ao_nn;
  }
  if (timing)
    for (int i = 0; i < 9; ++i) atomicAdd(&gstats[4 + i], wg_stat[i]);
}

// ---- B2: sort each row by column (mul_hash.rs:164-175) ----------------------------------------------
// Bitonic network over CAP LDS slots per row, padded with u32::MAX keys; G lanes per row.
template <class T, int G, int CAP>
__global__ __launch_bounds__((G > 256 ? G : 256)) void sort_rows_kernel(const uint32_t* __restrict__ rows_list,
                                                                       uint32_t nrows,
                                                                       const uint64_t* __restrict__ c_rp,
                                                                       uint32_t* __restrict__ c_col,
                                                                       T* __restrict__ c_val) {
  constexpr int TPB = (G > 256 ? G : 256);
  constexpr int RPW = TPB / G;
  __shared__ uint32_t sk[RPW][CAP];
  __shared__ T sv[RPW][CAP];
  const int g = threadIdx.x / G, gl = threadIdx.x % G;
  const uint32_t ridx = blockIdx.x * RPW + g;
  const bool valid = ridx < nrows;
  const uint32_t row = valid ? rows_list[ridx] : 0;
  const uint64_t s = valid ? c_rp[row] : 0;
  const uint32_t n = valid ? (uint32_t)(c_rp[row + 1] - s) : 0;
  // the network covers next_pow2(longest row of the workgroup) slots, and every lane takes whole compare-exchange
  // pairs (pair p of a stage with distance j: i = p with a zero bit inserted at j, partner i | j)
  uint32_t cap = next_pow2_u32(n < 2 ? 2 : n);
  if (RPW > 1) {
    __shared__ uint32_t cap_s;
    if (threadIdx.x == 0) cap_s = 0;
    __syncthreads();
    atomicMax(&cap_s, cap);
    __syncthreads();
    cap = cap_s;
  }
  for (uint32_t i = gl; i < cap; i += G) {
    sk[g][i] = i < n ? c_col[s + i] : EMPTY_KEY;
    sv[g][i] = i < n ? c_val[s + i] : T(0);
  }
  __syncthreads();
  for (uint32_t k = 2; k <= cap; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t p = gl; p < cap / 2; p += G) {
        const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
        const uint32_t ki = sk[g][i], kl = sk[g][l];
        const bool up = (i & k) == 0;
        if ((ki > kl) == up) {
          sk[g][i] = kl;
          sk[g][l] = ki;
          const T t = sv[g][i];
          sv[g][i] = sv[g][l];
          sv[g][l] = t;
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = gl; i < n; i += G) {
    c_col[s + i] = sk[g][i];
    c_val[s + i] = sv[g][i];
  }
}

// ---- reference slot order (sort_rows = 2) --------------------------------------------------------------
// The reference emits an unsorted row in linprobe SLOT order (map.rs:59-63), which depends on the order keys were
// first inserted (A-row storage order x B-row storage order).  Sequential linear probing in insertion order gives
// the same layout as the history-independent table in which an earlier key always takes precedence over a later one
// (every slot between a key's home and its position holds an earlier key), and that table can be built in parallel:
//   1. walk the row's product stream once more; a scratch table records, per distinct column, the smallest
//      stream position it occurs at (LDS atomicMin);
//   2. insert every column into the real table (capacity max(16, 2*next_pow2(nnz)), home (col*107) & mask) with a
//      64-bit LDS atomicMax on (~position, col): a higher-priority arrival evicts the resident, which moves on;
//   3. rank the occupied slots and rewrite the row of C (already computed, in any order) in that order.
// One 1024-thread workgroup per row, rows of up to 4096 entries (table <= 8192 slots = 128 KB of LDS in all).
template <class T>
__global__ __launch_bounds__(1024) void ref_order_kernel(const uint64_t* __restrict__ a_rp,
                                                        const uint32_t* __restrict__ a_col,
                                                        const uint64_t* __restrict__ b_rp,
                                                        const uint32_t* __restrict__ b_col,
                                                        const uint32_t* __restrict__ rows_list, uint32_t nrows,
                                                        const uint64_t* __restrict__ c_rp, uint32_t* __restrict__ c_col,
                                                        T* __restrict__ c_val) {
  constexpr int TPB = 1024, SEG = 1024, CAP = 8192, MAXN = 4096;
  __shared__ unsigned long long t2[CAP];                        // the reference-layout table: (~pos, col)
  __shared__ __attribute__((aligned(16))) uint32_t t1[2 * CAP];  // scratch: keys | minpos, later rank | staging
  __shared__ uint32_t seg_start[SEG + 1];
  __shared__ uint32_t seg_lo[SEG];
  __shared__ uint32_t wsum[TPB / 64 + 1];
  uint32_t* k1 = t1;
  uint32_t* mp = t1 + CAP;
  for (uint32_t ri = blockIdx.x; ri < nrows; ri += gridDim.x) {
    const uint32_t row = rows_list[ri];
    const uint64_t out0 = c_rp[row];
    const uint32_t n = (uint32_t)(c_rp[row + 1] - out0);
    const uint32_t cap = table_size_for(n), mask = cap - 1;
    for (uint32_t i = threadIdx.x; i < cap; i += TPB) {
      k1[i] = EMPTY_KEY;
      mp[i] = 0xFFFFFFFFu;
      t2[i] = 0ull;
    }
    __syncthreads();
    // 1. first-occurrence position of every distinct column
    const uint64_t ps = a_rp[row], pe = a_rp[row + 1];
    uint32_t done = 0;  // products in earlier batches
    for (uint64_t pb = ps; pb < pe; pb += SEG) {
      const uint64_t p = pb + threadIdx.x;
      uint32_t lo = 0, len = 0;
      if (p < pe) {
        const uint32_t k = a_col[p];
        lo = (uint32_t)b_rp[k];
        len = (uint32_t)b_rp[k + 1] - lo;
      }
      uint32_t total;
      const uint32_t st = block_excl_scan_tot<TPB>(len, wsum, total);
      seg_start[threadIdx.x] = st;
      seg_lo[threadIdx.x] = lo - st;  // q = seg_lo[s] + idx for a product index idx of segment s
      if (threadIdx.x == 0) seg_start[SEG] = total;
      __syncthreads();
      for (uint32_t idx = threadIdx.x; idx < total; idx += TPB) {
        const uint32_t sg = seg_find(seg_start, SEG, idx);
        const uint32_t key = b_col[seg_lo[sg] + idx];
        uint32_t h = (key * HASH_SCAL) & mask;
        for (;;) {
          const uint32_t old = atomicCAS(&k1[h], EMPTY_KEY, key);
          if (old == EMPTY_KEY || old == key) break;
          h = (h + 1) & mask;
        }
        atomicMin(&mp[h], done + idx);
      }
      done += total;
      __syncthreads();
    }
    // 2. priority insertion into the reference-layout table
    for (uint32_t i = threadIdx.x; i < cap; i += TPB) {
      const uint32_t key = k1[i];
      if (key == EMPTY_KEY) continue;
      unsigned long long v = ((unsigned long long)(~mp[i]) << 32) | key;
      uint32_t h = (key * HASH_SCAL) & mask;
      for (;;) {
        const unsigned long long old = atomicMax(&t2[h], v);
        if (old == 0ull) break;
        if (old < v) v = old;  // we evicted a later key: it carries on from the next slot
        h = (h + 1) & mask;
      }
    }
    __syncthreads();
    // 3. rank of every occupied slot = position in the drained row; t1 is free again
    uint16_t* rank = (uint16_t*)t1;               // [CAP]
    uint32_t* scol = t1 + CAP / 2;                // [MAXN]
    T* sval = (T*)(t1 + CAP / 2 + MAXN);          // [MAXN]
    const uint32_t spt = cap >= TPB ? cap / TPB : 1;
    uint32_t cnt = 0;
    if (threadIdx.x * spt < cap)
      for (uint32_t j = 0; j < spt; ++j) cnt += t2[threadIdx.x * spt + j] != 0ull;
    uint32_t r = block_excl_scan<TPB>(cnt, wsum);
    if (threadIdx.x * spt < cap)
      for (uint32_t j = 0; j < spt; ++j) {
        const uint32_t sl = threadIdx.x * spt + j;
        if (t2[sl] != 0ull) rank[sl] = (uint16_t)r++;
      }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      const uint32_t key = c_col[out0 + i];
      uint32_t h = (key * HASH_SCAL) & mask;
      while ((uint32_t)t2[h] != key || t2[h] == 0ull) h = (h + 1) & mask;
      const uint32_t rr = rank[h];
      scol[rr] = key;
      sval[rr] = c_val[out0 + i];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      c_col[out0 + i] = scol[i];
      c_val[out0 + i] = sval[i];
    }
    __syncthreads();
  }
}

// ---- reference slot order for rows longer than 4096 entries (tables in HBM) ---------------------------
// Same three steps as ref_order_kernel, but a 485 k-entry row needs the reference's 2^20-slot table, so:
//   1. minpos_heavy_kernel: first stream position of every distinct column, chunk by chunk through a dense
//      u32 array in LDS (2 numeric chunks = 32768 columns per pass, LDS atomicMin), written as (col, pos)
//      pairs in column order into scratch arrays laid out like C;
//   2. ref_order_long_kernel: one workgroup per row builds the priority table in its own HBM slab with 64-bit
//      global atomicMax, ranks the occupied slots and rewrites the row of C in that order.
// A verification path: global atomics and L1-bypassing loads throughout, not tuned.
constexpr uint32_t MP_COLS = 32768;  // columns per minpos pass (128 KB of u32)

template <bool USE_BPOS>
__global__ __launch_bounds__(1024) void minpos_heavy_kernel(
    const uint64_t* __restrict__ a_rp, const uint32_t* __restrict__ a_col, const uint64_t* __restrict__ b_rp,
    const uint32_t* __restrict__ b_col, const uint32_t* __restrict__ bpos, uint32_t rows_b, int cw_shift,
    const uint32_t* __restrict__ rows_list, uint32_t n_rows, uint32_t* __restrict__ queue,
    const uint32_t* __restrict__ hrow_of_row, uint32_t nchunks, const uint32_t* __restrict__ hoff,
    const uint64_t* __restrict__ c_rp, uint32_t* __restrict__ col2, uint32_t* __restrict__ pos2) {
  constexpr int TPB = 1024, SEG = 1024;
  __shared__ uint32_t mp[MP_COLS];
  __shared__ uint32_t seg_start[SEG + 1];
  __shared__ uint32_t seg_lo[SEG];
  __shared__ uint32_t seg_pos[SEG];  // stream position of the segment's first product
  __shared__ uint32_t wsum[TPB / 64 + 1];
  __shared__ uint32_t item_s;
  for (uint32_t i = threadIdx.x; i < MP_COLS; i += TPB) mp[i] = 0xFFFFFFFFu;
  const uint32_t cpg = MP_COLS >> cw_shift;  // numeric chunks per pass
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) item_s = atomicAdd(queue, 1u);
    __syncthreads();
    if (item_s >= n_rows) break;
    const uint32_t row = rows_list[item_s];
    const uint32_t hrow = hrow_of_row[row];
    const uint32_t* ho = hoff + (uint64_t)hrow * nchunks;
    const uint64_t ps = a_rp[row], pe = a_rp[row + 1];
    const uint64_t out_row = c_rp[row];
    for (uint32_t j0 = 0; j0 < nchunks; j0 += cpg) {
      const uint32_t j1 = j0 + cpg < nchunks ? j0 + cpg : nchunks;
      const uint32_t c0 = j0 << cw_shift;
      uint32_t carry = 0;  // products of the row's earlier A entries (all columns)
      for (uint64_t pb = ps; pb < pe; pb += SEG) {
        const uint64_t p = pb + threadIdx.x;
        uint32_t lo = 0, len = 0, full = 0, skip = 0;
        if (p < pe) {
          const uint32_t k = a_col[p];
          const uint32_t rs = (uint32_t)b_rp[k];
          full = (uint32_t)b_rp[k + 1] - rs;
          if (USE_BPOS) {
            lo = bpos[(uint64_t)j0 * rows_b + k];
            len = bpos[(uint64_t)j1 * rows_b + k] - lo;
          } else {
            lo = rs;
            len = full;
          }
          skip = lo - rs;
        }
        const uint32_t before = block_excl_scan<TPB>(full, wsum);
        __syncthreads();
        const uint32_t batch_full = wsum[TPB / 64];
        const uint32_t st = block_excl_scan<TPB>(len, wsum);
        seg_start[threadIdx.x] = st;
        seg_lo[threadIdx.x] = lo - st;  // q = seg_lo[s] + idx for a product index idx of segment s
        seg_pos[threadIdx.x] = carry + before + skip;
        __syncthreads();
        const uint32_t total = wsum[TPB / 64];
        if (threadIdx.x == 0) seg_start[SEG] = total;
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < total; idx += TPB) {
          const uint32_t sg = seg_find(seg_start, SEG, idx);
          const uint32_t off = idx - seg_start[sg];
          const uint32_t c = b_col[seg_lo[sg] + idx] - c0;
          if (c < MP_COLS && c < ((j1 - j0) << cw_shift)) atomicMin(&mp[c], seg_pos[sg] + off);
        }
        carry += batch_full;
        __syncthreads();
      }
      // drain in column order: thread t owns 32 consecutive columns; untouched = 0xFFFFFFFF
      uint32_t n = 0;
      for (int j = 0; j < (int)(MP_COLS / TPB); ++j) n += mp[threadIdx.x * (MP_COLS / TPB) + j] != 0xFFFFFFFFu;
      const uint32_t ex = block_excl_scan<TPB>(n, wsum);
      if (n) {
        uint64_t out = out_row + ho[j0] + ex;
        for (int j = 0; j < (int)(MP_COLS / TPB); ++j) {
          const uint32_t c = threadIdx.x * (MP_COLS / TPB) + j;
          const uint32_t v = mp[c];
          if (v != 0xFFFFFFFFu) {
            col2[out] = c0 + c;
            pos2[out] = v;
            mp[c] = 0xFFFFFFFFu;
            ++out;
          }
        }
      }
      __syncthreads();
    }
  }
}

template <class T>
__global__ __launch_bounds__(1024) void ref_order_long_kernel(const uint32_t* __restrict__ rows_list, uint32_t n_rows,
                                                             uint32_t* __restrict__ queue,
                                                             const uint64_t* __restrict__ c_rp,
                                                             const uint32_t* __restrict__ col2,
                                                             const uint32_t* __restrict__ pos2,
                                                             uint32_t* __restrict__ c_col, T* __restrict__ c_val,
                                                             unsigned long long* __restrict__ t2_all, uint64_t cap_max,
                                                             uint32_t* __restrict__ scol_all, T* __restrict__ sval_all,
                                                             uint64_t n_max) {
  constexpr int TPB = 1024;
  __shared__ uint32_t wsum[TPB / 64 + 1];
  __shared__ uint32_t item_s;
  unsigned long long* t2 = t2_all + (uint64_t)blockIdx.x * cap_max;
  uint32_t* scol = scol_all + (uint64_t)blockIdx.x * n_max;
  T* sval = sval_all + (uint64_t)blockIdx.x * n_max;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) item_s = atomicAdd(queue, 1u);
    __syncthreads();
    if (item_s >= n_rows) break;
    const uint32_t row = rows_list[item_s];
    const uint64_t out0 = c_rp[row];
    const uint32_t n = (uint32_t)(c_rp[row + 1] - out0);
    const uint32_t cap = table_size_for(n), mask = cap - 1;
    for (uint32_t i = threadIdx.x; i < cap; i += TPB) __hip_atomic_store(&t2[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __syncthreads();
    // priority insertion (see ref_order_kernel), global 64-bit atomicMax
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      const uint32_t key = col2[out0 + i];
      unsigned long long v = ((unsigned long long)(~pos2[out0 + i]) << 32) | key;
      uint32_t h = (key * HASH_SCAL) & mask;
      for (;;) {
        const unsigned long long old = atomicMax(&t2[h], v);
        if (old == 0ull) break;
        if (old < v) v = old;
        h = (h + 1) & mask;
      }
    }
    __threadfence();
    __syncthreads();
    // rank the occupied slots: thread t owns a contiguous range; slot becomes (rank + 1) << 32 | key
    const uint32_t per = (cap + TPB - 1) / TPB;
    const uint32_t s0 = threadIdx.x * per, s1 = s0 + per < cap ? s0 + per : cap;
    uint32_t cnt = 0;
    for (uint32_t sl = s0; sl < s1; ++sl)
      cnt += __hip_atomic_load(&t2[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    uint32_t r = block_excl_scan<TPB>(cnt, wsum);
    for (uint32_t sl = s0; sl < s1; ++sl) {
      const unsigned long long v = __hip_atomic_load(&t2[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != 0ull) {
        __hip_atomic_store(&t2[sl], ((unsigned long long)(r + 1) << 32) | (uint32_t)v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        ++r;
      }
    }
    __threadfence();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      const uint32_t key = c_col[out0 + i];
      uint32_t h = (key * HASH_SCAL) & mask;
      unsigned long long v;
      for (;;) {
        v = __hip_atomic_load(&t2[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 0ull && (uint32_t)v == key) break;
        h = (h + 1) & mask;
      }
      const uint32_t rr = (uint32_t)(v >> 32) - 1;
      __hip_atomic_store(&scol[rr], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sval[rr] = c_val[out0 + i];
    }
    __threadfence();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      c_col[out0 + i] = __hip_atomic_load(&scol[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c_val[out0 + i] = __builtin_nontemporal_load(&sval[i]);
    }
    __threadfence();
  }
}

// out[0] = longest row of the list; out[3] = 1 if a row has 2^32 or more intermediate products
__global__ __launch_bounds__(256) void max_nnz_kernel(const uint32_t* __restrict__ list, uint32_t n,
                                                     const uint32_t* __restrict__ row_nnz,
                                                     const uint64_t* __restrict__ flops, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t v = 0;
  if (i < n) {
    v = row_nnz[list[i]];
    if (flops[list[i]] >> 32) out[3] = 1;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d, 64));
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}

// ---- rows_to_threads' partition on the device (mul_hash.rs:51-62) ---------------------------------------
__global__ void part_cost_kernel(const uint64_t* __restrict__ flops, uint64_t rows, uint64_t row_overhead,
                                 uint64_t* __restrict__ cost) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) cost[i] = flops[i] + (flops[i] ? row_overhead : 0);
}
// offsets[t] = partition_point(ps, x <= average * t) - 1 for t = 1..nparts-1; offsets[0] = 0, offsets[nparts] = rows
__global__ void part_points_kernel(const uint64_t* __restrict__ ps, uint64_t rows, uint64_t nparts,
                                   uint64_t* __restrict__ offsets) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > nparts) return;
  if (t == 0) {
    offsets[0] = 0;
    return;
  }
  if (t == nparts) {
    offsets[t] = rows;
    return;
  }
  const uint64_t total = ps[rows];
  const uint64_t avg = total / nparts + (total % nparts != 0);
  const uint64_t bound = avg * t;
  uint64_t lo = 0, hi = rows + 1;  // first index with ps[idx] > bound
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (ps[mid] <= bound)
      lo = mid + 1;
    else
      hi = mid;
  }
  offsets[t] = lo - 1;
}

// ---- sort the rows of a CSR matrix by column (CsrMatrix<T, false> -> CsrMatrix<T, true>) -----------------
// Rows of up to 4096 entries: sort_rows_kernel (bitonic in LDS).  Longer rows: scatter into a dense 16384-column
// window (value + byte flag per column, entries of a row are distinct), drain the window in column order into a
// staging slab, copy back.  One workgroup per long row, cols / 16384 passes over the row.
__global__ __launch_bounds__(256) void row_len_kernel(const uint64_t* __restrict__ rp, uint32_t rows,
                                                     uint32_t* __restrict__ len) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) {
    const uint64_t l = rp[r + 1] - rp[r];
    len[r] = l > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l;
  }
}

template <class T>
__global__ __launch_bounds__(1024) void sort_long_rows_kernel(const uint32_t* __restrict__ rows_list, uint32_t n_rows,
                                                             uint32_t* __restrict__ queue,
                                                             const uint64_t* __restrict__ rp, uint32_t* __restrict__ col,
                                                             T* __restrict__ val, uint64_t ncols,
                                                             uint32_t* __restrict__ scol_all, T* __restrict__ sval_all,
                                                             uint64_t n_max) {
  constexpr int TPB = 1024;
  constexpr uint32_t W = 16384;
  __shared__ __attribute__((aligned(16))) T acc[W];
  __shared__ __attribute__((aligned(16))) uint8_t flags[W];
  __shared__ uint32_t wsum[TPB / 64 + 1];
  __shared__ uint32_t item_s;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4* f4 = (u32x4*)flags;
  f4[threadIdx.x] = (u32x4)(0u);
  uint32_t* scol = scol_all + (uint64_t)blockIdx.x * n_max;
  T* sval = sval_all + (uint64_t)blockIdx.x * n_max;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) item_s = atomicAdd(queue, 1u);
    __syncthreads();
    if (item_s >= n_rows) break;
    const uint32_t row = rows_list[item_s];
    const uint64_t s0 = rp[row];
    const uint32_t n = (uint32_t)(rp[row + 1] - s0);
    uint32_t done = 0;
    for (uint64_t c0 = 0; c0 < ncols && done < n; c0 += W) {
      for (uint32_t i = threadIdx.x; i < n; i += TPB) {
        const uint64_t c = (uint64_t)col[s0 + i] - c0;
        if (c < W) {
          acc[c] = val[s0 + i];
          flags[c] = 1;
        }
      }
      __syncthreads();
      const u32x4 w = f4[threadIdx.x];
      const uint32_t cnt = __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
      uint32_t tot;
      uint32_t out = done + block_excl_scan_tot<TPB>(cnt, wsum, tot);
      if (cnt) {
        f4[threadIdx.x] = (u32x4)(0u);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if ((ww[j >> 2] >> ((j & 3) * 8)) & 1u) {
            scol[out] = (uint32_t)(c0 + threadIdx.x * 16 + j);
            sval[out] = acc[threadIdx.x * 16 + j];
            ++out;
          }
      }
      done += tot;
      __syncthreads();
    }
    __threadfence();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      col[s0 + i] = __builtin_nontemporal_load(&scol[i]);
      val[s0 + i] = __builtin_nontemporal_load(&sval[i]);
    }
    __threadfence();
  }
}

// in place: hist[0 .. n) -> exclusive prefix, hist[n] = total.  One workgroup (n = 2 * chunks: a few hundred to a million).
__global__ __launch_bounds__(1024) void bucket_scan_kernel(uint32_t* __restrict__ hist, uint32_t n) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += hist[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 1024; ++i) {
      const uint32_t c = part[i];
      part[i] = run;
      run += c;
    }
    hist[n] = run;
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t c = hist[i];
    hist[i] = run;
    run += c;
  }
}

static int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
  }
  return cus;
}

}  // namespace spam

// ---- plan ---------------------------------------------------------------------------------------------
struct spam_spgemm_plan {
  const spam_csr* a = nullptr;
  const spam_csr* b = nullptr;
  spam::DevBuf flops, row_nnz, c_row_ptr, scan_tmp;
  spam::DevBuf heavy_rows;   // symbolic heavy list (copy of the last symbolic bin)
  spam::DevBuf hrow_of_row;  // u32[rows]: index into heavy_rows
  spam::DevBuf hcnt, hoff;   // u32[n_heavy * nchunks]
  spam::DevBuf bpos;         // u32[(nchunks + 1) * rows_b]
  spam::DevBuf bs_col, bs_val;  // row-sorted copy of an unsorted B (heavy path only)
  const uint32_t* b_col = nullptr;  // B's arrays as the heavy kernels see them
  const void* b_val = nullptr;
  uint32_t n_heavy = 0, nchunks = 0;
  int cw_shift = spam::CW_SHIFT_512;
  bool use_bpos = false;
  spam_spgemm_stats st{};
  hipEvent_t ev[6] = {};
  bool numeric_done = false;
};

namespace spam {

static void plan_destroy(spam_spgemm_plan* p) {
  if (!p) return;
  for (auto& e : p->ev)
    if (e) (void)hipEventDestroy(e);
  delete p;
}

static int launch_flops(const spam_csr* a, const spam_csr* b, uint64_t* d_flops, unsigned long long* d_total,
                        hipStream_t s) {
  SPAM_HIP_TRY(hipMemsetAsync(d_total, 0, 16, s));
  unsigned grid = (unsigned)std::min<uint64_t>((a->rows + 3) / 4, 16384);
  spgemm_flops_kernel<<<grid, 256, 0, s>>>(a->row_ptr, a->col_idx, b->row_ptr, (uint32_t)a->rows, d_flops, d_total);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

static int check_pair(const spam_csr* a, const spam_csr* b) {
  if (!a || !b) return set_error(SPAM_E_INVALID, "NULL matrix");
  if (a->cols != b->rows)
    return set_error(SPAM_E_INVALID,
                     "LHS cols (%llu) != RHS rows (%llu): the reference would panic indexing rhs.offsets[k + 1] "
                     "(mul_hash.rs:46)",
                     (unsigned long long)a->cols, (unsigned long long)b->rows);
  if (a->dtype != b->dtype) return set_error(SPAM_E_DTYPE, "A and B have different value types");
  if (b->nnz >= (1ull << 32))
    return set_error(SPAM_E_UNSUPPORTED, "nnz(B) >= 2^32: the SpGEMM kernels keep 32-bit positions into B's arrays");
  return SPAM_OK;
}

template <int G, int S, int TABLE>
static int launch_sym_hash(const spam_csr* a, const spam_csr* b, const uint64_t* flops, const uint32_t* list,
                           uint32_t n, uint32_t* row_nnz, hipStream_t s) {
  if (!n) return SPAM_OK;
  constexpr int TPB = (G > 256 ? G : 256);
  constexpr int RPW = TPB / G;
  sym_hash_kernel<G, S, TABLE><<<(n + RPW - 1) / RPW, TPB, 0, s>>>(a->row_ptr, a->col_idx, b->row_ptr, b->col_idx, flops,
                                                                    list, n, row_nnz);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T, int G, int S, int TABLE>
static int launch_num_hash(const spam_csr* a, const spam_csr* b, const uint32_t* list, uint32_t n,
                           const uint32_t* row_nnz, const uint64_t* c_rp, uint32_t* c_col, T* c_val, hipStream_t s) {
  if (!n) return SPAM_OK;
  constexpr int TPB = (G > 256 ? G : 256);
  constexpr int RPW = TPB / G;
  num_hash_kernel<T, G, S, TABLE><<<(n + RPW - 1) / RPW, TPB, 0, s>>>(a->row_ptr, a->col_idx, (const T*)a->vals,
                                                                       b->row_ptr, b->col_idx, (const T*)b->vals, list,
                                                                       n, row_nnz, c_rp, c_col, c_val);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

// out = list reordered heaviest-first by cost[row]; tmp64 is a zeroed u32[64] scratch
static int order_by_cost(const uint32_t* list, uint32_t n, const uint64_t* cost, uint32_t* tmp64, uint32_t* out,
                         hipStream_t s) {
  if (!n) return SPAM_OK;
  SPAM_HIP_TRY(hipMemsetAsync(tmp64, 0, 64 * 4, s));
  cost_hist_kernel<<<(n + 255) / 256, 256, 0, s>>>(list, n, cost, tmp64);
  SPAM_CHECK_LAUNCH();
  cost_scan_kernel<<<1, 64, 0, s>>>(tmp64);
  SPAM_CHECK_LAUNCH();
  cost_scatter_kernel<<<(n + 255) / 256, 256, 0, s>>>(list, n, cost, tmp64, out);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <int TPB, int TABLE>
static int launch_sym_row(const spam_csr* a, const spam_csr* b, const uint64_t* flops, const uint32_t* list, uint32_t n,
                          uint32_t* row_nnz, hipStream_t s) {
  if (!n) return SPAM_OK;
  sym_row_kernel<TPB, TABLE><<<std::min<uint32_t>(n, 16384), TPB, 0, s>>>(a->row_ptr, a->col_idx, b->row_ptr, b->col_idx,
                                                                        flops, list, n, row_nnz);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T, int TPB, int TABLE>
static int launch_num_row(const spam_csr* a, const spam_csr* b, const uint32_t* list, uint32_t n,
                          const uint32_t* row_nnz, const uint64_t* c_rp, uint32_t* c_col, T* c_val, hipStream_t s) {
  if (!n) return SPAM_OK;
  num_row_kernel<T, TPB, TABLE><<<std::min<uint32_t>(n, 16384), TPB, 0, s>>>(
      a->row_ptr, a->col_idx, (const T*)a->vals, b->row_ptr, b->col_idx, (const T*)b->vals, list, n, row_nnz, c_rp,
      c_col, c_val);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T, int G, int CAP>
static int launch_sort(const uint32_t* list, uint32_t n, const uint64_t* c_rp, uint32_t* c_col, T* c_val,
                       hipStream_t s) {
  if (!n) return SPAM_OK;
  constexpr int TPB = (G > 256 ? G : 256);
  constexpr int RPW = TPB / G;
  sort_rows_kernel<T, G, CAP><<<(n + RPW - 1) / RPW, TPB, 0, s>>>(list, n, c_rp, c_col, c_val);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

// Sort rows [in place] of arrays (rp, col, val).  `max_hint` 0 = unknown.
template <class T>
static int sort_rows_typed(uint64_t rows64, uint64_t cols, const uint64_t* rp, uint32_t* col, T* val, hipStream_t s) {
  const uint32_t rows = (uint32_t)rows64;
  DevBuf len, counts, lists, aux;
  SPAM_TRY(len.alloc((uint64_t)rows * 4));
  SPAM_TRY(counts.alloc(16 * 4));
  SPAM_TRY(lists.alloc((uint64_t)NNUM * rows * 4));
  SPAM_HIP_TRY(hipMemsetAsync(counts.p, 0, 64, s));
  row_len_kernel<<<(rows + 255) / 256, 256, 0, s>>>(rp, rows, len.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  BinLimits lim{};
  lim.nbins = NNUM;
  for (int i = 0; i < NNUM - 1; ++i) lim.max[i] = NUM_MAX[i];
  bin_rows_kernel<uint32_t><<<(rows + 255) / 256, 256, 0, s>>>(len.as<uint32_t>(), rows, lim, counts.as<uint32_t>(),
                                                              lists.as<uint32_t>(), nullptr);
  SPAM_CHECK_LAUNCH();
  uint32_t hc[8];
  SPAM_HIP_TRY(hipMemcpyAsync(hc, counts.p, 32, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  const uint32_t* L = lists.as<uint32_t>();
  SPAM_TRY((launch_sort<T, 16, 64>(L + 0ull * rows, hc[0], rp, col, val, s)));
  SPAM_TRY((launch_sort<T, 64, 512>(L + 1ull * rows, hc[1], rp, col, val, s)));
  SPAM_TRY((launch_sort<T, 256, 2048>(L + 2ull * rows, hc[2], rp, col, val, s)));
  SPAM_TRY((launch_sort<T, 1024, 4096>(L + 3ull * rows, hc[3], rp, col, val, s)));
  const uint32_t n_long = hc[NNUM - 1];
  if (n_long) {
    const uint32_t* hl = L + (uint64_t)(NNUM - 1) * rows;
    SPAM_TRY(aux.alloc(16));
    SPAM_HIP_TRY(hipMemsetAsync(aux.p, 0, 16, s));
    DevBuf zero64;
    SPAM_TRY(zero64.alloc((uint64_t)rows * 8));
    SPAM_HIP_TRY(hipMemsetAsync(zero64.p, 0, (uint64_t)rows * 8, s));  // "flops" all < 2^32 for max_nnz_kernel
    max_nnz_kernel<<<(n_long + 255) / 256, 256, 0, s>>>(hl, n_long, len.as<uint32_t>(), zero64.as<uint64_t>(), aux.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
    uint32_t nmax = 0;
    SPAM_HIP_TRY(hipMemcpyAsync(&nmax, aux.p, 4, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    const uint32_t nwg = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(n_long, 512),
                                                              (uint32_t)((8ull << 30) / ((uint64_t)nmax * (4 + sizeof(T))))));
    DevBuf scol, sval;
    SPAM_TRY(scol.alloc((uint64_t)nwg * nmax * 4));
    SPAM_TRY(sval.alloc((uint64_t)nwg * nmax * sizeof(T)));
    sort_long_rows_kernel<T><<<nwg, 1024, 0, s>>>(hl, n_long, aux.as<uint32_t>() + 1, rp, col, val, cols,
                                                  scol.as<uint32_t>(), sval.as<T>(), nmax);
    SPAM_CHECK_LAUNCH();
    SPAM_HIP_TRY(hipStreamSynchronize(s));
  }
  return SPAM_OK;
}

static int symbolic_impl(const spam_csr* a, const spam_csr* b, spam_spgemm_plan** plan_out, uint64_t* nnz_out) {
  SPAM_TRY(check_pair(a, b));
  hipStream_t s = stream();
  spam_spgemm_plan* p = new (std::nothrow) spam_spgemm_plan();
  if (!p) return set_error(SPAM_E_OOM, "host allocation failed");
  struct Guard {
    spam_spgemm_plan* p;
    ~Guard() { plan_destroy(p); }
  } guard{p};
  p->a = a, p->b = b;
  const uint32_t rows = (uint32_t)a->rows;
  spam_spgemm_stats& st = p->st;
  st.rows = a->rows, st.inner = a->cols, st.cols = b->cols, st.nnz_a = a->nnz, st.nnz_b = b->nnz;
  for (auto& e : p->ev) SPAM_HIP_TRY(hipEventCreate(&e));
  SPAM_HIP_TRY(hipEventRecord(p->ev[0], s));

  // L1 flop count
  DevBuf total;
  SPAM_TRY(p->flops.alloc((uint64_t)rows * 8));
  SPAM_TRY(total.alloc(16));
  SPAM_TRY(launch_flops(a, b, p->flops.as<uint64_t>(), total.as<unsigned long long>(), s));
  SPAM_HIP_TRY(hipEventRecord(p->ev[1], s));

  // bin rows by F
  DevBuf counts, lists;
  SPAM_TRY(counts.alloc(16 * 4));  // [0..7] bin counts, [8] work queue of sym_heavy_kernel
  SPAM_TRY(lists.alloc((uint64_t)NSYM * rows * 4));
  SPAM_TRY(p->hrow_of_row.alloc((uint64_t)rows * 4));
  SPAM_TRY(p->row_nnz.alloc((uint64_t)rows * 4));
  SPAM_HIP_TRY(hipMemsetAsync(counts.p, 0, 64, s));
  SPAM_HIP_TRY(hipMemsetAsync(p->row_nnz.p, 0, (uint64_t)rows * 4, s));  // rows with F = 0 are skipped (mul_hash.rs:84)
  BinLimits lim{};
  lim.nbins = NSYM;
  for (int i = 0; i < NSYM - 1; ++i) lim.max[i] = SYM_MAX[i];
  bin_rows_kernel<uint64_t><<<(rows + 255) / 256, 256, 0, s>>>(p->flops.as<uint64_t>(), rows, lim, counts.as<uint32_t>(),
                                                              lists.as<uint32_t>(), p->hrow_of_row.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  uint32_t hc[8];
  unsigned long long htot[2];
  SPAM_HIP_TRY(hipMemcpyAsync(hc, counts.p, 32, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(htot, total.p, 16, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (htot[1]) return set_error(SPAM_E_OVERFLOW, "intermediate-product count overflows usize (mul_hash.rs:47-48)");
  st.flops_f = htot[0];
  for (int i = 0; i < NSYM; ++i) st.rows_sym_bin[i] = hc[i];

  const uint32_t* L = lists.as<uint32_t>();
  uint32_t* rn = p->row_nnz.as<uint32_t>();
  const uint64_t* fl = p->flops.as<uint64_t>();
  SPAM_TRY((launch_sym_hash<16, 4, 128>(a, b, fl, L + 0ull * rows, hc[0], rn, s)));
  SPAM_TRY((launch_sym_hash<64, 8, 1024>(a, b, fl, L + 1ull * rows, hc[1], rn, s)));
  SPAM_TRY((launch_sym_row<256, 8192>(a, b, fl, L + 2ull * rows, hc[2], rn, s)));

  // heavy rows: dense flags per symbolic chunk
  p->n_heavy = hc[NSYM - 1];
  {
    const char* cfg = getenv("SPAM_HEAVY_TPB");  // 256 / 512 / 1024: force a heavy-path geometry (A/B runs)
    const int forced = cfg ? atoi(cfg) : 0;
    const uint64_t nch512 = (b->cols + (1u << CW_SHIFT_512) - 1) >> CW_SHIFT_512;
    p->cw_shift = forced == 256 ? CW_SHIFT_256 : forced == 1024 ? CW_SHIFT_1024 : forced == 512 ? CW_SHIFT_512
                  : (nch512 > WIDE_B_CHUNKS ? CW_SHIFT_1024 : CW_SHIFT_512);
    p->nchunks = (uint32_t)((b->cols + (1ull << p->cw_shift) - 1) >> p->cw_shift);
  }
  if (p->n_heavy) {
    SPAM_TRY(p->heavy_rows.alloc((uint64_t)p->n_heavy * 4));
    SPAM_HIP_TRY(hipMemcpyAsync(p->heavy_rows.p, L + (uint64_t)(NSYM - 1) * rows, (uint64_t)p->n_heavy * 4,
                                hipMemcpyDeviceToDevice, s));
    p->b_col = b->col_idx;
    p->b_val = b->vals;
    if (!b->is_sorted && p->nchunks > 1) {
      // the B pre-split needs sorted rows: sort a private copy once (the reference accepts either, flag B1)
      SPAM_TRY(wait_values(b, s));
      SPAM_TRY(p->bs_col.alloc(b->nnz * 4));
      SPAM_TRY(p->bs_val.alloc(b->nnz * dtype_size(b->dtype)));
      SPAM_HIP_TRY(hipMemcpyAsync(p->bs_col.p, b->col_idx, b->nnz * 4, hipMemcpyDeviceToDevice, s));
      SPAM_HIP_TRY(hipMemcpyAsync(p->bs_val.p, b->vals, b->nnz * dtype_size(b->dtype), hipMemcpyDeviceToDevice, s));
      SPAM_TRY(SPAM_DISPATCH_DTYPE(
          b->dtype, sort_rows_typed<double>(b->rows, b->cols, b->row_ptr, p->bs_col.as<uint32_t>(), p->bs_val.as<double>(), s),
          sort_rows_typed<float>(b->rows, b->cols, b->row_ptr, p->bs_col.as<uint32_t>(), p->bs_val.as<float>(), s),
          sort_rows_typed<i64w>(b->rows, b->cols, b->row_ptr, p->bs_col.as<uint32_t>(), p->bs_val.as<i64w>(), s)));
      p->b_col = p->bs_col.as<uint32_t>();
      p->b_val = p->bs_val.p;
    }
    p->use_bpos = p->nchunks > 1;
    if (p->use_bpos) {
      SPAM_TRY(p->bpos.alloc((uint64_t)(p->nchunks + 1) * b->rows * 4));
      // side array of the long rows' boundaries: a row of more than BPOS_SHORT entries holds at least BPOS_SHORT + 1
      DevBuf lslot, lb, nlong, litems;
      const uint64_t max_long = b->nnz / (BPOS_SHORT + 1) + 1, max_items = b->nnz / BPOS_SEG + max_long;
      const uint32_t lb_stride = (p->nchunks + 1 + 3) & ~3u;
      SPAM_TRY(lslot.alloc(b->rows * 4));
      SPAM_TRY(lb.alloc(max_long * lb_stride * 4 + 16));
      SPAM_TRY(litems.alloc(max_items * sizeof(BposItem)));
      SPAM_TRY(nlong.alloc(16));
      SPAM_HIP_TRY(hipMemsetAsync(nlong.p, 0, 16, s));
      bpos_long_items_kernel<<<(unsigned)((b->rows + 255) / 256), 256, 0, s>>>(b->row_ptr, (uint32_t)b->rows, nlong.as<uint32_t>(),
                                                                               lslot.as<uint32_t>(), litems.as<BposItem>());
      SPAM_CHECK_LAUNCH();
      bpos_long_kernel<<<(unsigned)((max_items + 3) / 4), 256, 0, s>>>(b->row_ptr, p->b_col, p->nchunks, lb_stride, p->cw_shift,
                                                                      nlong.as<uint32_t>(), lslot.as<uint32_t>(), litems.as<BposItem>(),
                                                                      lb.as<uint32_t>());
      SPAM_CHECK_LAUNCH();
      bpos_walk_kernel<<<(unsigned)((b->rows + 63) / 64), 64, 0, s>>>(b->row_ptr, p->b_col, (uint32_t)b->rows, p->nchunks, lb_stride,
                                                                      p->cw_shift, lslot.as<uint32_t>(), lb.as<uint32_t>(), p->bpos.as<uint32_t>());
      SPAM_CHECK_LAUNCH();
      // (lslot / lb go back to the allocator here; later users are queued on the same stream behind the walk kernel)
    }
    SPAM_TRY(p->hcnt.alloc((uint64_t)p->n_heavy * p->nchunks * 4));
    SPAM_TRY(p->hoff.alloc((uint64_t)p->n_heavy * p->nchunks * 4));
    DevBuf sym_order, otmp;
    SPAM_TRY(sym_order.alloc((uint64_t)p->n_heavy * 4));
    SPAM_TRY(otmp.alloc(64 * 4));
    SPAM_TRY(order_by_cost(p->heavy_rows.as<uint32_t>(), p->n_heavy, fl, otmp.as<uint32_t>(), sym_order.as<uint32_t>(), s));
    // geometry of the symbolic heavy kernel (SPAM_SYM_GEOM): 0 = 1024 threads, 2^20-column windows, one workgroup per
    // CU; 1 = 512 threads, 2^19 columns, two per CU; 2 = 1024 threads, 2^19 columns, two per CU (8 waves per SIMD); 3 = 512 threads, 2^18 columns, four per CU
    const char* sym_geom_cfg = getenv("SPAM_SYM_GEOM");  // (read per call: the tests run every geometry in one process)
    const int sym_geom_env = sym_geom_cfg ? atoi(sym_geom_cfg) : -1;
    // (scale-20 A*A: 28.7 / 27.8 / 24.8 / 24.2 ms of symbolic phase; every window costs two B-split lookups per entry of
    // A's row, so wide matrices take the larger windows)
    const int sym_geom = sym_geom_env >= 0 && sym_geom_env <= 3 ? sym_geom_env
                         : b->cols <= (16ull << 18) ? 3 : b->cols <= (32ull << 19) ? 2 : 0;
    const int win_shift = sym_geom == 0 ? 20 : sym_geom == 3 ? 18 : 19;
    const uint64_t nwin = (b->cols + (1ull << win_shift) - 1) >> win_shift;  // bitmap windows per row
    // rows that hold more than about 1/1024 of this call's products (at least 2^22; a power of two, so that they are the
    // first buckets of the cost order) are cut finer, into sub-windows of 1/32 window: one (row, window) item of theirs
    // would keep a workgroup busy for milliseconds after the rest of the chip has run dry — the rank blocks of a multi-GPU
    // product that hold the hubs felt it (symbolic phase of block 0 of 8 at scale 20: 7.9 -> 5.7 ms); a single-GPU product
    // of that size has enough other rows to hide them (threshold 2^25 there: a handful of rows)
    uint32_t n_giant = 0;
    int giant_log2 = 22;
    while (giant_log2 < 62 && (1ull << giant_log2) < st.flops_f / 1024) ++giant_log2;
    {
      const char* gs = getenv("SPAM_SYM_GIANT_SHIFT");  // A/B runs: move the threshold by powers of two
      if (gs) giant_log2 = std::max(12, std::min(62, giant_log2 + atoi(gs)));
    }
    const uint32_t sub_chunks = std::max<uint32_t>(1u, ((1u << win_shift) >> p->cw_shift) / 32u);
    if (p->use_bpos) {
      uint32_t ends[64];
      SPAM_HIP_TRY(hipMemcpyAsync(ends, otmp.p, sizeof ends, hipMemcpyDeviceToHost, s));  // bucket ends after the scatter
      SPAM_HIP_TRY(hipStreamSynchronize(s));
      n_giant = std::min<uint32_t>(ends[63 - giant_log2], p->n_heavy);
    }
    const uint64_t nsub = ((uint64_t)p->nchunks + sub_chunks - 1) / sub_chunks;
    const uint64_t n_sitems = (uint64_t)n_giant * nsub + (uint64_t)(p->n_heavy - n_giant) * nwin;
    if (n_sitems >= 0xFFFFFFFFull) return set_error(SPAM_E_UNSUPPORTED, "too many (heavy row, column window) pairs for the 32-bit work queue");
    DevBuf sitems;
    SPAM_TRY(sitems.alloc((uint64_t)p->n_heavy * sizeof(SymItem)));
    sym_items_kernel<<<(p->n_heavy + 255) / 256, 256, 0, s>>>(sym_order.as<uint32_t>(), p->n_heavy, a->row_ptr,
                                                            p->hrow_of_row.as<uint32_t>(), sitems.as<SymItem>());
    SPAM_CHECK_LAUNCH();
#define SPAM_SYM_HEAVY_G(BPOS_, TPB_, WS_, WPS_)                                                                     \
  do {                                                                                                               \
    auto kern = sym_heavy_kernel<BPOS_, TPB_, WS_, WPS_>;                                                            \
    int per_cu = 0;                                                                                                  \
    SPAM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, TPB_, 0));                              \
    if (per_cu < 1) per_cu = 1;                                                                                      \
    const unsigned sgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)per_cu * device_cus(), (n_sitems + ITEM_RUN - 1) / ITEM_RUN)); \
    if (dbg_sym) fprintf(stderr, "[spam dbg] symbolic heavy: %d threads, 2^%d columns, %d workgroups per CU, grid %u\n", TPB_, WS_, per_cu, sgrid); \
    kern<<<sgrid, TPB_, 0, s>>>(a->col_idx, b->row_ptr, p->b_col, BPOS_ ? p->bpos.as<uint32_t>() : nullptr, (uint32_t)b->rows, \
                                sitems.as<SymItem>(), p->n_heavy, (uint32_t)nwin, n_giant, sub_chunks, p->nchunks, p->cw_shift, \
                                p->hcnt.as<uint32_t>(), counts.as<uint32_t>() + 8);                                  \
  } while (0)
#define SPAM_SYM_HEAVY(BPOS_)                                                                                        \
  do {                                                                                                               \
    if (sym_geom == 0) SPAM_SYM_HEAVY_G(BPOS_, 1024, 20, 4);                                                         \
    else if (sym_geom == 1) SPAM_SYM_HEAVY_G(BPOS_, 512, 19, 4);                                                     \
    else if (sym_geom == 2) SPAM_SYM_HEAVY_G(BPOS_, 1024, 19, 8);                                                    \
    else SPAM_SYM_HEAVY_G(BPOS_, 512, 18, 8);                                                                        \
  } while (0)
    {
      static const bool dbg_sym = [] { const char* e = getenv("SPAM_DBG_HEAVY"); return e && (atoi(e) & 64); }();
      if (p->use_bpos) SPAM_SYM_HEAVY(true); else SPAM_SYM_HEAVY(false);
    }
#undef SPAM_SYM_HEAVY
#undef SPAM_SYM_HEAVY_G
    SPAM_CHECK_LAUNCH();
    heavy_finalize_kernel<<<(p->n_heavy + 3) / 4, 256, 0, s>>>(p->heavy_rows.as<uint32_t>(), p->n_heavy, p->nchunks,
                                                              p->hcnt.as<uint32_t>(), p->hoff.as<uint32_t>(), rn);
    SPAM_CHECK_LAUNCH();
  }

  // offsets = checked_inclusive_scan(row_nz) (mul_hash.rs:117)
  SPAM_TRY(p->c_row_ptr.alloc(((uint64_t)rows + 1) * 8));
  SPAM_TRY(p->scan_tmp.alloc(scan_tmp_bytes(rows)));
  SPAM_TRY(scan_u32_to_u64(rn, rows, p->c_row_ptr.as<uint64_t>(), p->scan_tmp.p, s));
  SPAM_HIP_TRY(hipEventRecord(p->ev[2], s));
  uint64_t nnz_c = 0;
  SPAM_HIP_TRY(hipMemcpyAsync(&nnz_c, p->c_row_ptr.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  st.nnz_c = nnz_c;
  (void)hipEventElapsedTime(&st.ms_flops, p->ev[0], p->ev[1]);
  (void)hipEventElapsedTime(&st.ms_symbolic, p->ev[1], p->ev[2]);
  if (nnz_out) *nnz_out = nnz_c;
  guard.p = nullptr;
  *plan_out = p;
  return SPAM_OK;
}

template <class T>
static int numeric_typed(spam_spgemm_plan* p, int sort_rows, spam_csr** c_out) {
  const spam_csr *a = p->a, *b = p->b;
  hipStream_t s = stream();
  const uint32_t rows = (uint32_t)a->rows;
  spam_spgemm_stats& st = p->st;
  SPAM_TRY(wait_values(a, s));  // operands gathered by spam_csr_allgather_rccl: the symbolic phase ran while their values
  SPAM_TRY(wait_values(b, s));  // were still in flight; the numeric phase is the first reader
  SPAM_HIP_TRY(hipEventRecord(p->ev[3], s));
  DevBuf ccol, cval;
  SPAM_TRY(ccol.alloc(st.nnz_c * 4));
  SPAM_TRY(cval.alloc(st.nnz_c * sizeof(T)));
  SPAM_HIP_TRY(hipEventRecord(p->ev[4], s));

  DevBuf counts, lists;
  SPAM_TRY(counts.alloc(8 * 4 + 8));
  SPAM_TRY(lists.alloc((uint64_t)NNUM * rows * 4));
  SPAM_HIP_TRY(hipMemsetAsync(counts.p, 0, 40, s));
  BinLimits lim{};
  lim.nbins = NNUM;
  for (int i = 0; i < NNUM - 1; ++i) lim.max[i] = NUM_MAX[i];
  bin_rows_kernel<uint32_t><<<(rows + 255) / 256, 256, 0, s>>>(p->row_nnz.as<uint32_t>(), rows, lim, counts.as<uint32_t>(),
                                                              lists.as<uint32_t>(), nullptr);
  SPAM_CHECK_LAUNCH();
  uint32_t hc[8];
  SPAM_HIP_TRY(hipMemcpyAsync(hc, counts.p, 32, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < NNUM; ++i) st.rows_num_bin[i] = hc[i];

  const uint32_t* L = lists.as<uint32_t>();
  const uint32_t* rn = p->row_nnz.as<uint32_t>();
  const uint64_t* crp = p->c_row_ptr.as<uint64_t>();
  uint32_t* cc = ccol.as<uint32_t>();
  T* cv = cval.as<T>();
  SPAM_TRY((launch_num_hash<T, 16, 4, 128>(a, b, L + 0ull * rows, hc[0], rn, crp, cc, cv, s)));
  SPAM_TRY((launch_num_hash<T, 64, 8, 1024>(a, b, L + 1ull * rows, hc[1], rn, crp, cc, cv, s)));
  SPAM_TRY((launch_num_row<T, 256, 4096>(a, b, L + 2ull * rows, hc[2], rn, crp, cc, cv, s)));
  SPAM_TRY((launch_num_row<T, 1024, 8192>(a, b, L + 3ull * rows, hc[3], rn, crp, cc, cv, s)));

  const uint32_t n_hnum = hc[NNUM - 1];
  if (n_hnum) {
    const char* dbg_env = getenv("SPAM_DBG_HEAVY");  // 32: print in-kernel phase timers (results unaffected)
    const int dbg_heavy = dbg_env ? atoi(dbg_env) : 0;
    const uint32_t* hl = L + (uint64_t)(NNUM - 1) * rows;
    // work items: one per (row, chunk group), bucketed by first chunk = the queue order
    const uint32_t nbuckets = p->nchunks;
    const int tpb = (1 << p->cw_shift) / 16;
    DevBuf gst, hist, items;
    SPAM_TRY(gst.alloc(256));
    SPAM_TRY(hist.alloc(((uint64_t)nbuckets + 2) * 4));
    SPAM_HIP_TRY(hipMemsetAsync(gst.p, 0, 256, s));
    SPAM_HIP_TRY(hipMemsetAsync(hist.p, 0, ((uint64_t)nbuckets + 2) * 4, s));
    unsigned long long* census = gst.as<unsigned long long>();           // [0] groups, [2] entries, [4..12] phase timers
    uint32_t* queue = (uint32_t*)(gst.as<unsigned long long>() + 16);    // work queue head
    const unsigned gg = (unsigned)std::min<uint64_t>(((uint64_t)n_hnum * 64 + 255) / 256, 2048);  // rows are strided over the waves
    const size_t gl = p->nchunks <= HIST_CAP ? (size_t)2 * p->nchunks * 4 : 0;
    heavy_groups_kernel<false><<<gg, 256, gl, s>>>(hl, n_hnum, p->hrow_of_row.as<uint32_t>(), p->nchunks, p->hcnt.as<uint32_t>(),
                                                  p->hoff.as<uint32_t>(), a->row_ptr, crp, (uint32_t)p->cw_shift, (uint32_t)sizeof(T),
                                                  heavy_region_bytes(tpb), hist.as<uint32_t>(), nullptr, census);
    SPAM_CHECK_LAUNCH();
    bucket_scan_kernel<<<1, 1024, 0, s>>>(hist.as<uint32_t>(), nbuckets);
    SPAM_CHECK_LAUNCH();
    uint32_t n_items = 0;
    SPAM_HIP_TRY(hipMemcpyAsync(&n_items, hist.as<uint32_t>() + nbuckets, 4, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    SPAM_TRY(items.alloc((uint64_t)std::max<uint32_t>(n_items, 1) * sizeof(HeavyItem)));
    heavy_groups_kernel<true><<<gg, 256, gl, s>>>(hl, n_hnum, p->hrow_of_row.as<uint32_t>(), p->nchunks, p->hcnt.as<uint32_t>(),
                                                 p->hoff.as<uint32_t>(), a->row_ptr, crp, (uint32_t)p->cw_shift, (uint32_t)sizeof(T),
                                                 heavy_region_bytes(tpb), hist.as<uint32_t>(), items.as<HeavyItem>(), census);
    SPAM_CHECK_LAUNCH();
#define SPAM_NUM_HEAVY_G(BPOS_, TPB_, CWS_)                                                                          \
  do {                                                                                                               \
    auto kern = num_heavy_kernel<T, BPOS_, TPB_, CWS_>;                                                              \
    int per_cu = 0;                                                                                                  \
    SPAM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, TPB_, 0));                              \
    if (per_cu < 1) per_cu = 1;                                                                                      \
    const uint64_t resident = (uint64_t)per_cu * (uint64_t)device_cus();                                             \
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(resident, (n_items + ITEM_RUN - 1) / ITEM_RUN)); \
    if (dbg_heavy & 64) fprintf(stderr, "[spam dbg] heavy numeric: %d threads, %d workgroups per CU, grid %u, %u items\n", TPB_, per_cu, grid, n_items); \
    kern<<<grid, TPB_, 0, s>>>(a->col_idx, (const T*)a->vals, b->row_ptr, p->b_col, (const T*)p->b_val,              \
                               BPOS_ ? p->bpos.as<uint32_t>() : nullptr, (uint32_t)b->rows, items.as<HeavyItem>(),   \
                               n_items, queue, cc, cv, census, dbg_heavy);                                           \
  } while (0)
#define SPAM_NUM_HEAVY(BPOS_)                                                               \
  do {                                                                                      \
    if (p->cw_shift == CW_SHIFT_256)                                                        \
      SPAM_NUM_HEAVY_G(BPOS_, 256, CW_SHIFT_256);                                           \
    else if (p->cw_shift == CW_SHIFT_512)                                                   \
      SPAM_NUM_HEAVY_G(BPOS_, 512, CW_SHIFT_512);                                           \
    else                                                                                    \
      SPAM_NUM_HEAVY_G(BPOS_, 1024, CW_SHIFT_1024);                                         \
  } while (0)
    if (n_items) {
      if (p->use_bpos) SPAM_NUM_HEAVY(true); else SPAM_NUM_HEAVY(false);
    }
#undef SPAM_NUM_HEAVY
#undef SPAM_NUM_HEAVY_G
    SPAM_CHECK_LAUNCH();
    st.heavy_items = n_hnum;
    unsigned long long hg[16];
    SPAM_HIP_TRY(hipMemcpyAsync(hg, gst.p, 128, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    if (dbg_heavy & 32) {
      const double ni = (double)(hg[12] ? hg[12] : 1);
      fprintf(stderr, "[spam dbg] heavy numeric, cycles per item (thread 0 of every workgroup): prefetch issue %.0f | pass 1 %.0f | barrier %.0f | "
              "prefix step %.0f | pass 2 %.0f | barrier %.0f | drain %.0f | closing barrier %.0f; %u rows, %u items\n", hg[4] / ni, hg[5] / ni,
              hg[6] / ni, hg[7] / ni, hg[8] / ni, hg[9] / ni, hg[10] / ni, hg[11] / ni, n_hnum, n_items);
    }
    st.heavy_dense_groups = hg[0], st.heavy_hash_groups = 0, st.heavy_dense_nnz = hg[2], st.heavy_hash_nnz = 0;
  }
  if (sort_rows == 2 && n_hnum) {
    // long rows: first-occurrence positions chunk by chunk, then the reference-layout table in HBM
    const uint32_t* hl = L + (uint64_t)(NNUM - 1) * rows;
    DevBuf aux, col2, pos2, t2, scol, sval;
    SPAM_TRY(aux.alloc(16));
    SPAM_HIP_TRY(hipMemsetAsync(aux.p, 0, 16, s));
    uint32_t* d_aux = aux.as<uint32_t>();  // [0] max nnz, [1] queue of pass 1, [2] queue of pass 2
    max_nnz_kernel<<<(n_hnum + 255) / 256, 256, 0, s>>>(hl, n_hnum, rn, p->flops.as<uint64_t>(), d_aux);
    SPAM_CHECK_LAUNCH();
    uint32_t haux[4] = {0, 0, 0, 0};
    SPAM_HIP_TRY(hipMemcpyAsync(haux, d_aux, 16, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    const uint32_t nmax = haux[0];
    if (haux[3])
      return set_error(SPAM_E_UNSUPPORTED, "reference slot order keeps 32-bit stream positions: a row has >= 2^32 products");
    uint64_t np2 = 1;
    while (np2 < nmax) np2 <<= 1;
    const uint64_t cap_max = std::max<uint64_t>(16, 2 * np2);  // the reference's capacity of the longest row
    const uint64_t per_wg = cap_max * 8 + (uint64_t)nmax * (4 + sizeof(T));
    const uint32_t nwg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(n_hnum, 512), (16ull << 30) / per_wg));
    SPAM_TRY(col2.alloc(st.nnz_c * 4));
    SPAM_TRY(pos2.alloc(st.nnz_c * 4));
    SPAM_TRY(t2.alloc((uint64_t)nwg * cap_max * 8));
    SPAM_TRY(scol.alloc((uint64_t)nwg * nmax * 4));
    SPAM_TRY(sval.alloc((uint64_t)nwg * nmax * sizeof(T)));
    const unsigned g1 = std::min<uint32_t>(n_hnum, HEAVY_GRID);
    if (p->use_bpos && b->is_sorted)  // stream positions follow B's OWN storage order
      minpos_heavy_kernel<true><<<g1, 1024, 0, s>>>(a->row_ptr, a->col_idx, b->row_ptr, b->col_idx, p->bpos.as<uint32_t>(),
                                                    (uint32_t)b->rows, p->cw_shift, hl, n_hnum, d_aux + 1,
                                                    p->hrow_of_row.as<uint32_t>(), p->nchunks, p->hoff.as<uint32_t>(), crp,
                                                    col2.as<uint32_t>(), pos2.as<uint32_t>());
    else
      minpos_heavy_kernel<false><<<g1, 1024, 0, s>>>(a->row_ptr, a->col_idx, b->row_ptr, b->col_idx, nullptr,
                                                     (uint32_t)b->rows, p->cw_shift, hl, n_hnum, d_aux + 1,
                                                     p->hrow_of_row.as<uint32_t>(), p->nchunks, p->hoff.as<uint32_t>(), crp,
                                                     col2.as<uint32_t>(), pos2.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
    ref_order_long_kernel<T><<<nwg, 1024, 0, s>>>(hl, n_hnum, d_aux + 2, crp, col2.as<uint32_t>(), pos2.as<uint32_t>(), cc, cv,
                                                  t2.as<unsigned long long>(), cap_max, scol.as<uint32_t>(), sval.as<T>(), nmax);
    SPAM_CHECK_LAUNCH();
    SPAM_HIP_TRY(hipStreamSynchronize(s));  // scratch goes back to the pool at scope exit
  }
  if (sort_rows == 2) {
    // rows of at most 4096 entries: everything in LDS
    for (int bin = 0; bin < NNUM - 1; ++bin) {
      if (!hc[bin]) continue;
      ref_order_kernel<T><<<std::min<uint32_t>(hc[bin], 8192), 1024, 0, s>>>(
          a->row_ptr, a->col_idx, b->row_ptr, b->col_idx, L + (uint64_t)bin * rows, hc[bin], crp, cc, cv);
      SPAM_CHECK_LAUNCH();
    }
  } else if (sort_rows) {
    SPAM_TRY((launch_sort<T, 16, 64>(L + 0ull * rows, hc[0], crp, cc, cv, s)));
    SPAM_TRY((launch_sort<T, 64, 512>(L + 1ull * rows, hc[1], crp, cc, cv, s)));
    SPAM_TRY((launch_sort<T, 256, 2048>(L + 2ull * rows, hc[2], crp, cc, cv, s)));
    SPAM_TRY((launch_sort<T, 1024, 4096>(L + 3ull * rows, hc[3], crp, cc, cv, s)));
  }
  SPAM_HIP_TRY(hipEventRecord(p->ev[5], s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  (void)hipEventElapsedTime(&st.ms_alloc, p->ev[3], p->ev[4]);
  (void)hipEventElapsedTime(&st.ms_numeric, p->ev[4], p->ev[5]);
  st.ms_total = st.ms_flops + st.ms_symbolic + st.ms_alloc + st.ms_numeric;

  spam_csr* c = new (std::nothrow) spam_csr();
  if (!c) return set_error(SPAM_E_OOM, "host allocation failed");
  c->rows = a->rows, c->cols = b->cols, c->nnz = st.nnz_c, c->dtype = a->dtype;
  // heavy rows are always sorted; hash rows only when asked (the const generic B2)
  c->is_sorted = sort_rows == 1 ? 1 : 0;
  c->row_ptr = (uint64_t*)p->c_row_ptr.release();
  c->col_idx = (uint32_t*)ccol.release();
  c->vals = cval.release();
  c->owns_row_ptr = c->owns_col_val = true;
  if (c->nnz < (1ull << 32)) {
    DevBuf r32;
    if (int e = r32.alloc((c->rows + 1) * 4)) {
      spam_csr_free(c);
      return e;
    }
    c->row_ptr32 = (uint32_t*)r32.release();
    c->owns_row_ptr32 = true;
    if (int e = narrow_row_ptr(c->row_ptr, c->rows + 1, c->row_ptr32, s)) {
      spam_csr_free(c);
      return e;
    }
  }
  p->numeric_done = true;
  *c_out = c;
  return SPAM_OK;
}

static int numeric_dev_impl(spam_spgemm_plan* p, int sort_rows, spam_csr** c_out) {
  if (!p || !c_out) return set_error(SPAM_E_INVALID, "NULL argument");
  *c_out = nullptr;
  if (p->numeric_done) return set_error(SPAM_E_INVALID, "plan already consumed by a numeric call");
  return SPAM_DISPATCH_DTYPE(p->a->dtype, numeric_typed<double>(p, sort_rows, c_out), numeric_typed<float>(p, sort_rows, c_out),
                             numeric_typed<i64w>(p, sort_rows, c_out));
}

}  // namespace spam

using namespace spam;

extern "C" {

int spam_spgemm_symbolic(const spam_csr* a, const spam_csr* b, spam_spgemm_plan** plan, uint64_t* nnz_out) {
  clear_error();
  if (!plan) return set_error(SPAM_E_INVALID, "plan is NULL");
  *plan = nullptr;
  return symbolic_impl(a, b, plan, nnz_out);
}

int spam_spgemm_numeric_dev(spam_spgemm_plan* plan, int sort_rows, spam_csr** c_out) {
  clear_error();
  return numeric_dev_impl(plan, sort_rows, c_out);
}

int spam_spgemm_numeric_f64(spam_spgemm_plan* plan, int sort_rows, uint64_t* offsets_out, uint64_t* indices_out,
                            double* vals_out) {
  clear_error();
  if (plan && plan->a->dtype != SPAM_F64) return set_error(SPAM_E_DTYPE, "plan does not hold f64 matrices");
  spam_csr* c = nullptr;
  SPAM_TRY(numeric_dev_impl(plan, sort_rows, &c));
  int e = spam_csr_download_f64(c, offsets_out, indices_out, vals_out);
  spam_csr_free(c);
  return e;
}

int spam_spgemm_numeric_f32(spam_spgemm_plan* plan, int sort_rows, uint64_t* offsets_out, uint64_t* indices_out,
                            float* vals_out) {
  clear_error();
  if (plan && plan->a->dtype != SPAM_F32) return set_error(SPAM_E_DTYPE, "plan does not hold f32 matrices");
  spam_csr* c = nullptr;
  SPAM_TRY(numeric_dev_impl(plan, sort_rows, &c));
  int e = spam_csr_download_f32(c, offsets_out, indices_out, vals_out);
  spam_csr_free(c);
  return e;
}

int spam_spgemm_numeric_i64(spam_spgemm_plan* plan, int sort_rows, uint64_t* offsets_out, uint64_t* indices_out,
                            int64_t* vals_out) {
  clear_error();
  if (plan && plan->a->dtype != SPAM_I64) return set_error(SPAM_E_DTYPE, "plan does not hold i64 matrices");
  spam_csr* c = nullptr;
  SPAM_TRY(numeric_dev_impl(plan, sort_rows, &c));
  int e = spam_csr_download_i64(c, offsets_out, indices_out, vals_out);
  spam_csr_free(c);
  return e;
}

int spam_csr_sort_rows(spam_csr* m) {
  clear_error();
  if (!m) return set_error(SPAM_E_INVALID, "matrix is NULL");
  if (m->is_sorted) return SPAM_OK;
  if (!m->owns_col_val) return set_error(SPAM_E_INVALID, "cannot sort a borrowed matrix in place (views, spam_csr_from_device)");
  hipStream_t s = stream();
  SPAM_TRY(wait_values(m, s));
  SPAM_TRY(SPAM_DISPATCH_DTYPE(m->dtype, sort_rows_typed<double>(m->rows, m->cols, m->row_ptr, m->col_idx, (double*)m->vals, s),
                               sort_rows_typed<float>(m->rows, m->cols, m->row_ptr, m->col_idx, (float*)m->vals, s),
                               sort_rows_typed<i64w>(m->rows, m->cols, m->row_ptr, m->col_idx, (i64w*)m->vals, s)));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  m->is_sorted = 1;
  if (m->spmv_plan) {  // the hot-column remap mirrors col_idx
    spmv_plan_free(m->spmv_plan);
    m->spmv_plan = nullptr;
  }
  return SPAM_OK;
}

int spam_spgemm_plan_stats(const spam_spgemm_plan* plan, spam_spgemm_stats* stats) {
  clear_error();
  if (!plan || !stats) return set_error(SPAM_E_INVALID, "NULL argument");
  *stats = plan->st;
  return SPAM_OK;
}

void spam_spgemm_plan_free(spam_spgemm_plan* plan) { plan_destroy(plan); }

int spam_spgemm_dev(const spam_csr* a, const spam_csr* b, int sort_rows, spam_csr** c_out, spam_spgemm_stats* stats) {
  clear_error();
  if (!c_out) return set_error(SPAM_E_INVALID, "c_out is NULL");
  *c_out = nullptr;
  spam_spgemm_plan* p = nullptr;
  uint64_t nnz = 0;
  SPAM_TRY(symbolic_impl(a, b, &p, &nnz));
  int e = numeric_dev_impl(p, sort_rows, c_out);
  if (stats) *stats = p->st;
  plan_destroy(p);
  return e;
}

// rows_to_threads' partition (mul_hash.rs:51-62) of a device vector of per-row product counts
static int parts_from_flops(const uint64_t* d_flops, uint64_t rows, uint64_t nparts, uint64_t row_overhead,
                            uint64_t* rows_offset_out, hipStream_t s) {
  if (row_overhead >= (1ull << 30)) return set_error(SPAM_E_OVERFLOW, "partition cost would overflow the u64 prefix sums (lib.rs:270)");
  DevBuf cost, ps, tmp, offs;
  SPAM_TRY(cost.alloc(rows * 8));
  SPAM_TRY(ps.alloc((rows + 1) * 8));
  SPAM_TRY(tmp.alloc(scan_tmp_bytes(rows)));
  SPAM_TRY(offs.alloc((nparts + 1) * 8));
  part_cost_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, s>>>(d_flops, rows, row_overhead, cost.as<uint64_t>());
  SPAM_CHECK_LAUNCH();
  SPAM_TRY(scan_u64_to_u64(cost.as<uint64_t>(), rows, ps.as<uint64_t>(), tmp.p, s));
  part_points_kernel<<<(unsigned)((nparts + 1 + 63) / 64), 64, 0, s>>>(ps.as<uint64_t>(), rows, nparts, offs.as<uint64_t>());
  SPAM_CHECK_LAUNCH();
  SPAM_HIP_TRY(hipMemcpyAsync(rows_offset_out, offs.p, (nparts + 1) * 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  return SPAM_OK;
}

static int rows_to_parts_impl(const spam_csr* a, const spam_csr* b, uint64_t nparts, uint64_t row_overhead,
                              uint64_t* row_flops_out, uint64_t* rows_offset_out) {
  clear_error();
  if (!rows_offset_out || nparts == 0) return set_error(SPAM_E_INVALID, "bad nparts / NULL output");
  SPAM_TRY(check_pair(a, b));
  if (row_overhead == SPAM_ROW_OVERHEAD_AUTO) row_overhead = spam_row_overhead_auto(b);
  hipStream_t s = stream();
  DevBuf flops, total;
  SPAM_TRY(flops.alloc(a->rows * 8));
  SPAM_TRY(total.alloc(16));
  SPAM_TRY(launch_flops(a, b, flops.as<uint64_t>(), total.as<unsigned long long>(), s));
  unsigned long long htot[2];
  SPAM_HIP_TRY(hipMemcpyAsync(htot, total.p, 16, hipMemcpyDeviceToHost, s));
  if (row_flops_out) SPAM_HIP_TRY(hipMemcpyAsync(row_flops_out, flops.p, a->rows * 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (htot[1]) return set_error(SPAM_E_OVERFLOW, "intermediate-product count overflows usize (mul_hash.rs:47-48)");
  // the scanned cost adds row_overhead per non-empty row to F: keep the u64 prefix sums of checked_inclusive_scan
  // (lib.rs:267-274) from wrapping (rows < 2^32, so F < 2^63 leaves room for any sane overhead)
  if (htot[0] >= (1ull << 63)) return set_error(SPAM_E_OVERFLOW, "partition cost would overflow the u64 prefix sums (lib.rs:270)");
  return parts_from_flops(flops.as<uint64_t>(), a->rows, nparts, row_overhead, rows_offset_out, s);
}

// The per-row term of the partition cost, in products.  A non-empty row of the heavy path costs work in proportion to
// the number of column chunks it is cut along (B-split lookups, work items, drain), so the term grows with the width
// of B: 32 products per 8192-column chunk — 4000 at scale 20 (128 chunks), where 0 / 2300 / 4000 / 6000 gave mean / max
// block times of 0.91 / 0.94 / 0.96 / 0.95, and 16 000 at scale 22 (512 chunks), where 4000 / 8000 / 12 000 / 16 000 gave
// 0.81 / 0.85 / 0.89 / see profiles/r04_partition_balance.txt.
uint64_t spam_row_overhead_auto(const spam_csr* b) {
  if (!b) return 4000;
  const uint64_t chunks = (b->cols + (1ull << CW_SHIFT_512) - 1) >> CW_SHIFT_512;
  return std::max<uint64_t>(1000, std::min<uint64_t>(32 * chunks, 1ull << 20));
}

int spam_row_flops_dev(const spam_csr* a, const spam_csr* b, uint64_t row_lo, uint64_t row_hi, uint64_t* d_flops_out) {
  clear_error();
  SPAM_TRY(check_pair(a, b));
  if (row_lo > row_hi || row_hi > a->rows || (row_hi > row_lo && !d_flops_out))
    return set_error(SPAM_E_INVALID, "row range [%llu, %llu) outside 0..%llu or NULL output", (unsigned long long)row_lo,
                     (unsigned long long)row_hi, (unsigned long long)a->rows);
  if (row_hi == row_lo) return SPAM_OK;
  hipStream_t s = stream();
  DevBuf total;
  SPAM_TRY(total.alloc(16));
  SPAM_HIP_TRY(hipMemsetAsync(total.p, 0, 16, s));
  const uint64_t n = row_hi - row_lo;
  unsigned grid = (unsigned)std::min<uint64_t>((n + 3) / 4, 16384);
  spgemm_flops_kernel<<<grid, 256, 0, s>>>(a->row_ptr + row_lo, a->col_idx, b->row_ptr, (uint32_t)n, d_flops_out,
                                           total.as<unsigned long long>());
  SPAM_CHECK_LAUNCH();
  unsigned long long htot[2];
  SPAM_HIP_TRY(hipMemcpyAsync(htot, total.p, 16, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (htot[1] || htot[0] >= (1ull << 60))
    return set_error(SPAM_E_OVERFLOW, "intermediate-product count overflows usize (mul_hash.rs:47-48)");
  return SPAM_OK;
}

int spam_rows_to_parts_from_flops_dev(const uint64_t* d_row_flops, uint64_t rows, uint64_t nparts, uint64_t row_overhead,
                                      uint64_t* rows_offset_out) {
  clear_error();
  if (!d_row_flops || !rows_offset_out || nparts == 0 || rows == 0) return set_error(SPAM_E_INVALID, "bad nparts / rows / NULL argument");
  return parts_from_flops(d_row_flops, rows, nparts, row_overhead, rows_offset_out, stream());
}

int spam_spgemm_rank_step(const spam_csr* a_full, const spam_csr* b_full, uint64_t rank, uint64_t world,
                          uint64_t row_overhead, int sort_rows, spam_csr** c_block, uint64_t* row_lo, uint64_t* row_hi,
                          spam_spgemm_stats* stats) {
  clear_error();
  if (!c_block || !row_lo || !row_hi) return set_error(SPAM_E_INVALID, "NULL output");
  *c_block = nullptr;
  if (world == 0 || rank >= world) return set_error(SPAM_E_INVALID, "rank %llu outside world %llu", (unsigned long long)rank, (unsigned long long)world);
  std::vector<uint64_t> offs(world + 1);
  SPAM_TRY(rows_to_parts_impl(a_full, b_full, world, row_overhead, nullptr, offs.data()));
  *row_lo = offs[rank], *row_hi = offs[rank + 1];
  if (stats) *stats = spam_spgemm_stats{};
  if (*row_hi <= *row_lo) return SPAM_OK;  // an empty block (more ranks than non-empty rows): *c_block stays NULL
  spam_csr* blk = nullptr;
  SPAM_TRY(spam_csr_row_block(a_full, *row_lo, *row_hi, &blk));
  const int e = spam_spgemm_dev(blk, b_full, sort_rows, c_block, stats);
  spam_csr_free(blk);
  return e;
}

int spam_rows_to_parts(const spam_csr* a, const spam_csr* b, uint64_t nparts, uint64_t* row_flops_out,
                       uint64_t* rows_offset_out) {
  return rows_to_parts_impl(a, b, nparts, 0, row_flops_out, rows_offset_out);
}

int spam_rows_to_parts_weighted(const spam_csr* a, const spam_csr* b, uint64_t nparts, uint64_t row_overhead,
                                uint64_t* rows_offset_out) {
  return rows_to_parts_impl(a, b, nparts, row_overhead, nullptr, rows_offset_out);
}

}  // extern "C"
