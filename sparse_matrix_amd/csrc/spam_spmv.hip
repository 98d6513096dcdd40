// spam_spmv.hip — CSR SpMV y = A x for gfx950.
//
// Reference counterpart: none — the reference has no SpMV (its only product is mul_hash,
// spam_csr/src/lib.rs:292-297).  The result is DEFINED as A.mul_hash(X) with X = x as an n x 1 matrix
// (mul_hash.rs:145-163 with the single key 0): y_i = sum over row i in storage order,
// y_i = 0 for rows without entries.
//
// Design (HBM-bound: 12 B/nnz of ja/a stream + the x gather; no MFMA, nothing here is a dense contraction):
//   * the nnz range is cut into fixed tiles of TPB*NPT entries, so every workgroup streams the same number of
//     bytes whatever the row-length distribution (R-MAT, 50k-entry rows next to 8-entry rows) — this is the row
//     "binning" of CSR-adaptive done per tile;
//   * tile_issue: 16-byte coalesced non-temporal loads of ja / a, the x gathers, and behind them the tile's slice
//     of row_ptr; tile_finish: products into an LDS tile (padded one slot per 32), row offsets into LDS;
//   * tile_phase2, from LDS only: rows of <= 32 entries are summed by one lane each, left to right (bit-identical
//     to the oracle's order), rows of <= 256 entries by 16 lanes, longer rows and row pieces by a wavefront (DPP sums);
//   * a row spanning several tiles gets its pieces from `carry[tile]`, folded in tile order by a second tiny
//     kernel — deterministic, no float atomics.
//   The row -> tile map (first row starting in each tile) depends on row_ptr only and is built once per
//   matrix (the analysis step of CSR-adaptive), then reused by every SpMV on that handle.
//
// Plain path: spmv_tile_kernel, one workgroup per tile, optional hot-column table (one 3 MB table in every L2).
//
// XCD-sliced path (second half of this file): when x is larger than one XCD's 4 MB L2 a random gather that
// misses L2 moves a whole 128-byte line over the fabric (52 G gathers/s against 245 G/s for L2 hits,
// profiles/r01_gather_microbench.txt).  MI355X has EIGHT such L2s; tools/xcd_gather_bench.hip shows they serve
// 230 G gathers/s from a 24-28 MB table as long as every workgroup only touches the table slice of the XCD it
// runs on.  So the analysis deals the most referenced columns over 8 slabs (one packed x-table per XCD), stores
// the rows of >= long_min entries again as 8 slab CSRs + a cold part, and the persistent, software-pipelined
// spmv_sliced_pipe_kernel has its workgroups read HW_REG_XCC_ID and pull tiles of their own XCD's slab from a
// per-XCD queue (then the cold part, then what other XCDs have left, so nothing depends on the placement).  That is
// the TILE form, kept as a deterministic alternative (SPAM_SPMV_RG=0); by default the sliced rows are held in ROW-GROUP
// form (spam_spmv_rg.hip: accumulators in LDS, entries in table-slot order), which is also a candidate where x fits an L2.
// Every part yields a partial y over the long rows; sliced_reduce_kernel adds them in part order.  Shorter rows
// stay in a compact residual CSR (spmv_short_rows_kernel).  The first SpMV on a handle times both paths on the
// caller's x and keeps the faster one.
//
// Environment (read per call; a change rebuilds the plan):  SPAM_SPMV_SLICED 0 never / 1 decide (default) / 2 always;
// SPAM_SPMV_SLICE_KB table bytes per XCD (3072);  SPAM_SPMV_LDS_KB hottest slots copied to LDS (-1 decide, 0 never);
// SPAM_SPMV_L0_KB contiguous hottest level without LDS (16);  SPAM_SPMV_LONG rows with at least this many entries go
// to the slabs (0: from the one-table miss rate);  SPAM_SPMV_SLICED_TPB / _NPT / _WGS kernel geometry;  SPAM_SPMV_PIPE
// 0 plain loop / 1 pipelined (default);  SPAM_SPMV_PROF 1 in-kernel phase timers, 2 skip phase 2 (diagnostics);
// plain path: SPAM_SPMV_NPT, SPAM_SPMV_NT, SPAM_SPMV_HOT, SPAM_SPMV_HOT_KB.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "spam_internal.hpp"

namespace spam {

struct SlicedPlan;
static void sliced_plan_free(SlicedPlan* q);

// Tuning knobs, read from the environment ONCE per plan (spam_spmv_prepare / the first SpMV on a handle).
struct SpmvKnobs {
  int npt = 0;           // SPAM_SPMV_NPT        entries per thread of the plain tile kernel (8 / 16; 0: by matrix size)
  int nt = 1;            // SPAM_SPMV_NT         non-temporal ja / a stream
  int hot = 2;           // SPAM_SPMV_HOT        0 off, 1 per-lane base select, 2 cold gathers non-temporal
  int hot_kb = 3072;     // SPAM_SPMV_HOT_KB     hot-column table budget (of the 4 MB L2 per XCD)
  int sliced = 1;        // SPAM_SPMV_SLICED     0 never, 1 decide (timed in spam_spmv_prepare), 2 always
  int slice_kb = 3072;   // SPAM_SPMV_SLICE_KB   table bytes per XCD
  int lds_kb = -1;       // SPAM_SPMV_LDS_KB     hottest slots copied to LDS (-1 decide, 0 never)
  int l0_kb = 16;        // SPAM_SPMV_L0_KB      contiguous hottest level without LDS
  int long_min = 0;      // SPAM_SPMV_LONG       rows with at least this many entries go to the slabs (0: model)
  int tpb = 256, snpt = 8, wgs = 16, pipe = 1;  // SPAM_SPMV_SLICED_TPB / _NPT / _WGS, SPAM_SPMV_PIPE
  int prof = 0;          // SPAM_SPMV_PROF       1 in-kernel phase timers, 2 skip phase 2 (diagnostics, wrong results)
  int rg = 1;            // SPAM_SPMV_RG         long rows of the sliced path in row-group form (LDS accumulators,
                         //                      spam_spmv_rg.hip); 0: the tile form
  RgKnobs rgk;           // SPAM_SPMV_RG_{EBLK,NT,K,ABITS}
};

struct SpmvPlan {
  std::mutex mu;                    // serialises plan construction (first SpMV / prepare / values_changed) on a handle
  bool prepared = false;            // spam_spmv_prepare ran: hot table and XCD-sliced structure were analysed
  SpmvKnobs knobs;
  uint32_t* first_owned = nullptr;  // [ntiles + 1]: first row whose start lies in tile t (rows for t = ntiles)
  void* carry = nullptr;            // T[ntiles]
  uint64_t ntiles = 0;
  int npt = 0;                      // nnz per thread the map was built for
  // hot-column table (analysis): the hot_k most referenced columns get slots in a small dense table xh
  // that stays L2-resident; ja_hot[p] = HOT_BIT | slot for those, the plain column otherwise.
  int hot_kb = -1;                  // table budget the analysis ran for (-1: not analysed)
  uint32_t hot_k = 0;               // 0: disabled (x fits L2 anyway, or the hot set covers too little)
  double hot_cover = 0.0;           // fraction of nnz whose column is hot
  uint32_t* ja_hot = nullptr;       // u32[nnz]
  uint32_t* hot_cols = nullptr;     // u32[hot_k]
  void* xh = nullptr;               // T[hot_k], rebuilt from x by every SpMV
  // XCD-sliced mode (analysis + one timing comparison on the first SpMV)
  SlicedPlan* sliced = nullptr;     // non-null: SpMV runs sliced
  float ms_plain = 0, ms_sliced = 0;  // the timed comparison of spam_spmv_prepare (0: not run)
};

void spmv_plan_free(SpmvPlan* p) {
  if (!p) return;
  if (p->first_owned) dev_free(p->first_owned);
  if (p->carry) dev_free(p->carry);
  if (p->ja_hot) dev_free(p->ja_hot);
  if (p->hot_cols) dev_free(p->hot_cols);
  if (p->xh) dev_free(p->xh);
  if (p->sliced) sliced_plan_free(p->sliced);
  delete p;
}

constexpr uint32_t HOT_BIT = 0x80000000u;

constexpr int SPMV_TPB = 256;
constexpr int SHORT_ROW = 32;

template <class OffT>
__global__ void spmv_first_owned_kernel(const OffT* __restrict__ row_ptr, uint32_t rows, uint32_t tile,
                                        uint64_t ntiles, uint32_t* __restrict__ fo) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  if (t == ntiles) {
    fo[t] = rows;
    return;
  }
  const uint64_t target = t * tile;
  uint32_t lo = 0, hi = rows;  // first r in [0, rows] with row_ptr[r] >= target  (row_ptr[rows] = nnz > target)
  while (lo < hi) {
    uint32_t mid = lo + ((hi - lo) >> 1);
    if ((uint64_t)row_ptr[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  fo[t] = lo;
}

template <class T>
struct Vec4;
template <>
struct Vec4<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  static constexpr int N = 2;
};
template <>
struct Vec4<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  static constexpr int N = 4;
};
template <>
struct Vec4<i64w> {
  typedef i64w type __attribute__((ext_vector_type(2)));
  static constexpr int N = 2;
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Explicit global address space: pointers that a kernel reads from a descriptor in memory are generic to the
// compiler (flat_load, which also occupies the LDS counter and is waited for one at a time); the tile body takes
// global pointers so both kernels get global_load / global_store.
#define SPAM_GLB __attribute__((address_space(1)))

__device__ __forceinline__ int pad_idx(int i) { return i + (i >> 5); }

// Sum over the wavefront through DPP moves (no LDS crossbar traffic, ~10x less latency than six ds_bpermute
// rounds for a double); the total arrives in lane 63 only.  Lanes without a source lane add 0.
template <int CTRL, int ROW_MASK, class T>
__device__ __forceinline__ T dpp_shift(T v) {
  if constexpr (sizeof(T) == 8) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(T, ((long long)hi << 32) | (long long)(unsigned)lo);
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
  }
}
// Sum over each 16-lane row of the wavefront: the total of lanes 16g .. 16g+15 arrives in lane 16g+15.
template <class T>
__device__ __forceinline__ T row16_sum_lane15(T v) {
  v += dpp_shift<0x111, 0xF>(v);
  v += dpp_shift<0x112, 0xF>(v);
  v += dpp_shift<0x114, 0xF>(v);
  v += dpp_shift<0x118, 0xF>(v);
  return v;
}
template <class T>
__device__ __forceinline__ T wave_sum_lane63(T v) {
  v += dpp_shift<0x111, 0xF>(v);  // row_shr:1
  v += dpp_shift<0x112, 0xF>(v);  // row_shr:2
  v += dpp_shift<0x114, 0xF>(v);  // row_shr:4
  v += dpp_shift<0x118, 0xF>(v);  // row_shr:8  -> lane 15 of every 16-lane row holds the row's sum
  v += dpp_shift<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_shift<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds everything
  return v;
}

// HOT: 0 = col holds plain column indices; 1 = col is ja_hot, one load with a per-lane base (xh or x);
//      2 = col is ja_hot, hot entries from xh, cold entries by a non-temporal load (no L2 allocation).
template <class T, int HOT>
__device__ __forceinline__ T gather_x(const T* __restrict__ x, const T* __restrict__ xh, uint32_t c) {
  if (HOT == 0) return x[c];
  const uint32_t i = c & ~HOT_BIT;
  if (HOT == 1) {
    const T* b = (c & HOT_BIT) ? xh : x;
    return b[i];
  }
  if (c & HOT_BIT) return xh[i];
  return __builtin_nontemporal_load(x + i);
}

template <class T, int HOT>
struct GatherPlain {
  const T* __restrict__ x;
  const T* __restrict__ xh;
  __device__ __forceinline__ T operator()(uint32_t c) const { return gather_x<T, HOT>(x, xh, c); }
};

template <class T, int TPB, int NPT>
struct TileSmem {
  static constexpr int TILE = TPB * NPT;
  static constexpr int MAXLONG = TILE / (SHORT_ROW + 1) + 2;
  static constexpr int RCAP = TILE >= 4096 ? 2048 : 1024;  // row_ptr values staged per pass (tile-local offsets fit 16 bits)
  static constexpr int RK = RCAP / TPB;        // of which every thread prefetches RK while the gathers are in flight
  static_assert(TILE <= 65535 && RCAP % TPB == 0, "tile-local offsets are kept as u16");
  T prod[TILE + TILE / 32 + 1];
  uint16_t rp[RCAP];
  static constexpr int MID_ROW = 256;          // rows of SHORT_ROW+1 .. MID_ROW entries: 16 lanes each, longer: a wavefront
  static constexpr int MAXBIG = TILE / (MID_ROW + 1) + 2;
  uint32_t long_list[MAXLONG];                 // mid rows
  uint32_t big_list[MAXBIG];
  uint32_t n_long, n_big;
};

// In-kernel phase timers (SPAM_SPMV_PROF=1, sliced kernel only; results unaffected): cycles of wave 0 summed over
// all tiles: [0] stream + gather + products, [1] barrier, [2] short rows, [3] barrier + long rows, [4] dequeue, [5] tiles
__device__ unsigned long long g_spmv_prof[8];

// Everything one thread holds of a FULL tile between issuing its loads and writing the products to LDS.
template <class T, class OffT, int TPB, int NPT>
struct TileRegs {
  using VT = typename Vec4<T>::type;
  static constexpr int VN = Vec4<T>::N;     // values per 16-byte load
  static constexpr int NV = NPT / 4;        // 16-byte index loads per thread
  static constexpr int RK = TileSmem<T, TPB, NPT>::RK;
  VT v[NV][4 / VN];
  T xv[NV][4];
  OffT rraw[RK];
  uint32_t f0, f1;
};

// Full tile, step 1: issue every global load of the tile — ja / a stream, the x gathers, and behind them the tile's
// slice of row_ptr (unconditional loads with a clamped index: they go out back to back; the compiler barriers keep
// them from being hoisted in front of the stream and waited for one at a time).  Nothing is waited for beyond what
// the gather addresses need.
template <class T, class OffT, int TPB, int NPT, bool NT, class G>
__device__ __forceinline__ void tile_issue(TileRegs<T, OffT, TPB, NPT>& r, const OffT SPAM_GLB* __restrict__ row_ptr,
                                           const uint32_t SPAM_GLB* __restrict__ col, const T SPAM_GLB* __restrict__ val,
                                           const G& gx, const uint32_t SPAM_GLB* __restrict__ fo, uint32_t tile_idx) {
  using R = TileRegs<T, OffT, TPB, NPT>;
  using VT = typename R::VT;
  constexpr int VN = R::VN, NV = R::NV, RK = R::RK, TILE = TPB * NPT;
  const int tid = threadIdx.x;
  const uint64_t base = (uint64_t)tile_idx * TILE;
  r.f0 = fo[tile_idx], r.f1 = fo[tile_idx + 1];
  u32x4 c[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const uint64_t q = base + (uint64_t)(i * TPB + tid) * 4;
    if (NT) {
      c[i] = __builtin_nontemporal_load((const SPAM_GLB u32x4*)(col + q));
#pragma unroll
      for (int j = 0; j < 4 / VN; ++j) r.v[i][j] = __builtin_nontemporal_load((const SPAM_GLB VT*)(val + q) + j);
    } else {
      c[i] = *(const SPAM_GLB u32x4*)(col + q);
#pragma unroll
      for (int j = 0; j < 4 / VN; ++j) r.v[i][j] = *((const SPAM_GLB VT*)(val + q) + j);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r.xv[i][j] = gx(c[i][j]);
  }
  asm volatile("" ::: "memory");
  const uint32_t nv0 = r.f1 - r.f0 + 1;
#pragma unroll
  for (int k = 0; k < RK; ++k) {
    const uint32_t i = (uint32_t)(k * TPB + tid);
    r.rraw[k] = row_ptr[r.f0 + (i < nv0 ? i : nv0 - 1)];
  }
  asm volatile("" ::: "memory");
}

// Full tile, step 2: products and the tile-local row offsets into LDS (waits for the loads of tile_issue).
template <class T, class OffT, int TPB, int NPT>
__device__ __forceinline__ void tile_finish(const TileRegs<T, OffT, TPB, NPT>& r, TileSmem<T, TPB, NPT>& sm,
                                            uint32_t tile_idx) {
  using R = TileRegs<T, OffT, TPB, NPT>;
  constexpr int VN = R::VN, NV = R::NV, RK = R::RK, TILE = TPB * NPT;
  const int tid = threadIdx.x;
  const uint64_t base = (uint64_t)tile_idx * TILE, end = base + TILE;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = (i * TPB + tid) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) sm.prod[pad_idx(q + j)] = r.v[i][j / VN][j % VN] * r.xv[i][j];
  }
#pragma unroll
  for (int k = 0; k < RK; ++k) {
    const uint64_t rv = (uint64_t)r.rraw[k];
    sm.rp[k * TPB + tid] = (uint16_t)((rv < end ? rv : end) - base);
  }
}

// Step 3 (after a barrier): per-row sums of the tile's `cnt` products, everything from LDS.  sm.rp holds the first
// min(nv, RCAP) tile-local offsets of rows f0 .. f1; tiles with more rows (runs of empty or one-entry rows) restage.
// sm.n_long and sm.n_big must be 0 on entry.
template <class T, class OffT, int TPB, int NPT, int PROF = 0>
__device__ __forceinline__ void tile_phase2(TileSmem<T, TPB, NPT>& sm, const OffT SPAM_GLB* __restrict__ row_ptr,
                                            T SPAM_GLB* __restrict__ y, T SPAM_GLB* __restrict__ carry, uint32_t f0,
                                            uint32_t f1, uint64_t base, uint64_t end, uint32_t tile_idx, long long* pt3) {
  using SM = TileSmem<T, TPB, NPT>;
  constexpr int RCAP = SM::RCAP;
  constexpr uint32_t CARRY = 0xFFFFFFFFu;
  T* prod = sm.prod;
  uint32_t* long_list = sm.long_list;
  const int tid = threadIdx.x;
  const uint32_t nv = f1 - f0 + 1;  // row_ptr values of this tile: rows f0 .. f1 - 1 plus the closing offset
  const uint32_t cnt1 = (uint32_t)(end - base);
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t row0 = 0;                // rows finished in earlier passes
  for (;;) {
    const uint32_t n = (nv - row0) < (uint32_t)RCAP ? (nv - row0) : (uint32_t)RCAP;  // staged values rp[0 .. n)
    const uint32_t nr = n - 1;                                                       // complete rows of this pass
    // row f0-1 started in an earlier tile: its piece [0, rp[0]) goes to carry[tile]
    if (row0 == 0 && tid == 0 && f0 > 0 && sm.rp[0] > 0) {
      if (sm.rp[0] > (uint32_t)SM::MID_ROW)
        sm.big_list[atomicAdd(&sm.n_big, 1u)] = CARRY;
      else
        long_list[atomicAdd(&sm.n_long, 1u)] = CARRY;
    }
    // short rows: one lane per row, two rows per lane and round so that their LDS reads overlap; each row is summed
    // left to right (the oracle's order), the first four products of both rows are read together
    for (uint32_t i = tid; i < nr; i += 2 * TPB) {
      const uint32_t i2 = i + TPB;
      const bool has2 = i2 < nr;
      const uint32_t s1 = sm.rp[i], e1 = sm.rp[i + 1];
      const uint32_t s2 = has2 ? sm.rp[i2] : 0u, e2 = has2 ? sm.rp[i2 + 1] : 0u;
      const uint32_t len1 = e1 - s1, len2 = e2 - s2;
      const uint32_t last = cnt1 - 1;  // any valid product slot for the clamped reads
      T p[4], q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        p[k] = prod[pad_idx((int)(s1 + k < e1 ? s1 + k : last))];
        q[k] = prod[pad_idx((int)(s2 + k < e2 ? s2 + k : last))];
      }
      if (len1 <= SHORT_ROW) {
        T acc = T(0);
        if (len1) {
          acc = p[0];
#pragma unroll
          for (int k = 1; k < 4; ++k)
            if ((uint32_t)k < len1) acc += p[k];
          uint32_t k = 4;
          for (; k + 4 <= len1; k += 4) {
            const T p0 = prod[pad_idx((int)(s1 + k))], p1 = prod[pad_idx((int)(s1 + k + 1))];
            const T p2 = prod[pad_idx((int)(s1 + k + 2))], p3 = prod[pad_idx((int)(s1 + k + 3))];
            acc += p0;
            acc += p1;
            acc += p2;
            acc += p3;
          }
          for (; k < len1; ++k) acc += prod[pad_idx((int)(s1 + k))];
        }
        y[f0 + row0 + i] = acc;
      } else if (len1 <= (uint32_t)SM::MID_ROW) {
        long_list[atomicAdd(&sm.n_long, 1u)] = i;
      } else {
        sm.big_list[atomicAdd(&sm.n_big, 1u)] = i;
      }
      if (has2) {
        if (len2 <= SHORT_ROW) {
          T acc = T(0);
          if (len2) {
            acc = q[0];
#pragma unroll
            for (int k = 1; k < 4; ++k)
              if ((uint32_t)k < len2) acc += q[k];
            uint32_t k = 4;
            for (; k + 4 <= len2; k += 4) {
              const T p0 = prod[pad_idx((int)(s2 + k))], p1 = prod[pad_idx((int)(s2 + k + 1))];
              const T p2 = prod[pad_idx((int)(s2 + k + 2))], p3 = prod[pad_idx((int)(s2 + k + 3))];
              acc += p0;
              acc += p1;
              acc += p2;
              acc += p3;
            }
            for (; k < len2; ++k) acc += prod[pad_idx((int)(s2 + k))];
          }
          y[f0 + row0 + i2] = acc;
        } else if (len2 <= (uint32_t)SM::MID_ROW) {
          long_list[atomicAdd(&sm.n_long, 1u)] = i2;
        } else {
          sm.big_list[atomicAdd(&sm.n_big, 1u)] = i2;
        }
      }
    }
    if (PROF && row0 == 0) *pt3 = clock64();
    __syncthreads();
    // mid rows (SHORT_ROW+1 .. MID_ROW entries): four at a time per wavefront, 16 lanes each
    {
      const uint32_t nm = sm.n_long;
      const int grp = lane >> 4, sub = lane & 15;
      for (uint32_t jb = (uint32_t)wave * 4; jb < nm; jb += (TPB / 64) * 4) {
        const uint32_t j = jb + grp;
        const bool valid = j < nm;
        const uint32_t i = valid ? long_list[j] : 0u;
        uint32_t s = 0, e = 0;
        if (valid) {
          if (i == CARRY) {
            e = sm.rp[0];
          } else {
            s = sm.rp[i];
            e = sm.rp[i + 1];
          }
        }
        T acc = T(0);
        for (int pp = (int)s + sub, pe = (int)e; pp < pe; pp += 16) acc += prod[pad_idx(pp)];
        acc = row16_sum_lane15(acc);
        if (valid && sub == 15) {
          if (i == CARRY)
            carry[tile_idx] = acc;
          else
            y[f0 + row0 + i] = acc;
        }
      }
    }
    // big rows: one wavefront each
    const uint32_t nl = sm.n_big;
    for (uint32_t j = wave; j < nl; j += TPB / 64) {
      const uint32_t i = sm.big_list[j];
      uint32_t s = 0, e;
      if (i == CARRY) {
        e = sm.rp[0];
      } else {
        s = sm.rp[i];
        e = sm.rp[i + 1];
      }
      T acc = T(0);
      int p = (int)s + lane;
      const int pe = (int)e;
      for (; p + 192 < pe; p += 256) {
        const T p0 = prod[pad_idx(p)], p1 = prod[pad_idx(p + 64)], p2 = prod[pad_idx(p + 128)], p3 = prod[pad_idx(p + 192)];
        acc += (p0 + p1) + (p2 + p3);
      }
      for (; p < pe; p += 64) acc += prod[pad_idx(p)];
      acc = wave_sum_lane63(acc);
      if (lane == 63) {
        if (i == CARRY)
          carry[tile_idx] = acc;
        else
          y[f0 + row0 + i] = acc;
      }
    }
    row0 += nr;
    if (row0 + 1 >= nv) break;
    __syncthreads();  // everyone is done with rp / long_list of this pass
    if (tid == 0) sm.n_long = 0, sm.n_big = 0;
    const uint32_t n2 = (nv - row0) < (uint32_t)RCAP ? (nv - row0) : (uint32_t)RCAP;
    for (uint32_t i = tid; i < n2; i += TPB) {
      const uint64_t rv = (uint64_t)row_ptr[f0 + row0 + i];
      sm.rp[i] = (uint16_t)((rv < end ? rv : end) - base);
    }
    __syncthreads();
  }
}

// One tile of TPB*NPT entries start to end: products into LDS, then per-row sums.  `tile_idx` is the tile's number
// inside the (row_ptr, col, val) arrays it is given; G maps a stored column code to the x value.
// Phase 2 never waits on global memory: the tile's slice of row_ptr (rows fo[t] .. fo[t+1]) is loaded while the
// gathers are in flight and staged in LDS as tile-local offsets.
template <class T, class OffT, int TPB, int NPT, bool NT, class G, int PROF = 0>
__device__ __forceinline__ void spmv_tile_body(TileSmem<T, TPB, NPT>& sm, const OffT SPAM_GLB* __restrict__ row_ptr,
                                               const uint32_t SPAM_GLB* __restrict__ col,
                                               const T SPAM_GLB* __restrict__ val, const G& gx, T SPAM_GLB* __restrict__ y,
                                               const uint32_t SPAM_GLB* __restrict__ fo, T SPAM_GLB* __restrict__ carry,
                                               uint64_t nnz, uint32_t tile_idx) {
  using SM = TileSmem<T, TPB, NPT>;
  constexpr int TILE = TPB * NPT;
  constexpr int RK = SM::RK;
  const int tid = threadIdx.x;
  const uint64_t base = (uint64_t)tile_idx * TILE;
  const uint32_t cnt = (uint32_t)((nnz - base) < (uint64_t)TILE ? (nnz - base) : (uint64_t)TILE);
  const uint64_t end = base + cnt;
  if (tid == 0) sm.n_long = 0, sm.n_big = 0;
  long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0;
  if (PROF) pt0 = clock64();
  uint32_t f0, f1;
  if (cnt == TILE) {
    TileRegs<T, OffT, TPB, NPT> r;
    tile_issue<T, OffT, TPB, NPT, NT>(r, row_ptr, col, val, gx, fo, tile_idx);
    tile_finish<T, OffT, TPB, NPT>(r, sm, tile_idx);
    f0 = r.f0, f1 = r.f1;
  } else {
    f0 = fo[tile_idx], f1 = fo[tile_idx + 1];
    const uint32_t nv0 = f1 - f0 + 1;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      const uint32_t i = (uint32_t)(k * TPB + tid);
      const uint64_t rv = (uint64_t)row_ptr[f0 + (i < nv0 ? i : nv0 - 1)];
      sm.rp[k * TPB + tid] = (uint16_t)((rv < end ? rv : end) - base);
    }
    for (uint32_t q = tid; q < cnt; q += TPB) sm.prod[pad_idx(q)] = val[base + q] * gx(col[base + q]);
  }
  if (PROF == 2) return;  // experiment: phase 1 only (wrong results)
  if (PROF) pt1 = clock64();
  __syncthreads();
  if (PROF) pt2 = clock64();
  tile_phase2<T, OffT, TPB, NPT, PROF>(sm, row_ptr, y, carry, f0, f1, base, end, tile_idx, &pt3);
  if (PROF && tid == 0) {
    const long long pt4 = clock64();
    atomicAdd(&g_spmv_prof[0], (unsigned long long)(pt1 - pt0));
    atomicAdd(&g_spmv_prof[1], (unsigned long long)(pt2 - pt1));
    atomicAdd(&g_spmv_prof[2], (unsigned long long)(pt3 - pt2));
    atomicAdd(&g_spmv_prof[3], (unsigned long long)(pt4 - pt3));
    atomicAdd(&g_spmv_prof[5], 1ull);
  }
}

template <class T, class OffT, int NPT, bool NT, int HOT>
__global__ __launch_bounds__(SPMV_TPB) void spmv_tile_kernel(const OffT* __restrict__ row_ptr,
                                                            const uint32_t* __restrict__ col,
                                                            const T* __restrict__ val, const T* __restrict__ x,
                                                            const T* __restrict__ xh, T* __restrict__ y,
                                                            const uint32_t* __restrict__ fo, T* __restrict__ carry,
                                                            uint64_t nnz) {
  __shared__ TileSmem<T, SPMV_TPB, NPT> sm;
  const GatherPlain<T, HOT> gx{x, xh};
  spmv_tile_body<T, OffT, SPMV_TPB, NPT, NT>(sm, (const OffT SPAM_GLB*)row_ptr, (const uint32_t SPAM_GLB*)col,
                                             (const T SPAM_GLB*)val, gx, (T SPAM_GLB*)y, (const uint32_t SPAM_GLB*)fo,
                                             (T SPAM_GLB*)carry, nnz, blockIdx.x);
}

// Fold the pieces of rows that span several tiles, in tile order.
template <class T, class OffT>
__global__ void spmv_fixup_kernel(const OffT* __restrict__ row_ptr, const uint32_t* __restrict__ fo,
                                  const T* __restrict__ carry, T* __restrict__ y, uint64_t ntiles, uint32_t tile) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 || t >= ntiles) return;
  const uint32_t f = fo[t];
  if (f == 0) return;
  const uint64_t rs = row_ptr[f];
  if (!(rs > t * tile)) return;   // no row continues into tile t
  if (fo[t - 1] == f) return;     // tile t-1 lies inside the same row: its chain head does the work
  T acc = y[f - 1];
  for (uint64_t u = t; u < ntiles && fo[u] == f && rs > u * tile; ++u) acc += carry[u];
  y[f - 1] = acc;
}

template <class T>
__global__ void fill_zero_kernel(T* __restrict__ y, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = T(0);
}
template <class T>
__global__ void fill_const_kernel(T* __restrict__ y, uint64_t n, T v) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = v;
}

// ---- hot-column analysis (once per matrix) ----------------------------------------------------------
__global__ void col_count_kernel(const uint32_t* __restrict__ col, uint64_t nnz, uint32_t* __restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) atomicAdd(&cnt[col[i]], 1u);
}
// The same histogram without atomics (round 4): the column indices sorted (radix sort on the bits a column needs), the
// counts read off the run boundaries.  162.7 M device-scope atomic adds on 10 M counters with R-MAT's skew took 12.7 ms —
// the hot columns' counters serialise, and an agent-scope atomic on a chip of eight L2s is a trip to the memory side;
// three radix passes over 0.65 GB and one boundary pass take a third of that.  Used from 4 M entries on.
__global__ void col_runs_kernel(const uint32_t* __restrict__ sorted_col, uint64_t nnz, uint32_t* __restrict__ first,
                                uint32_t* __restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    const uint32_t c = sorted_col[i];
    if (i == 0 || sorted_col[i - 1] != c) first[c] = (uint32_t)i;                // (nnz < 2^32: checked by the caller)
  }
}
__global__ void col_runs_end_kernel(const uint32_t* __restrict__ sorted_col, uint64_t nnz, const uint32_t* __restrict__ first,
                                    uint32_t* __restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    const uint32_t c = sorted_col[i];
    if (i + 1 == nnz || sorted_col[i + 1] != c) cnt[c] = (uint32_t)(i + 1) - first[c];
  }
}
// out[0] = #columns with cnt >= t, out[1] = their total count
__global__ void count_ge_kernel(const uint32_t* __restrict__ cnt, uint32_t cols, uint32_t t,
                                unsigned long long* __restrict__ out) {
  unsigned long long n = 0, m = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cols; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = cnt[i];
    if (c >= t) {
      n += 1;
      m += c;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    n += __shfl_xor(n, d, 64);
    m += __shfl_xor(m, d, 64);
  }
  if ((threadIdx.x & 63) == 0 && n) {
    atomicAdd(&out[0], n);
    atomicAdd(&out[1], m);
  }
}
__global__ void hot_flag_kernel(const uint32_t* __restrict__ cnt, uint32_t cols, uint32_t t, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cols) flag[i] = cnt[i] >= t;
}
// slot[c] = HOT_BIT | rank for hot columns, c otherwise; hot_cols[rank] = c
__global__ void hot_assign_kernel(const uint32_t* __restrict__ flag, const uint64_t* __restrict__ pos, uint32_t cols,
                                  uint32_t* __restrict__ slot, uint32_t* __restrict__ hot_cols) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols) return;
  if (flag[i]) {
    const uint32_t r = (uint32_t)pos[i];
    slot[i] = HOT_BIT | r;
    hot_cols[r] = (uint32_t)i;
  } else {
    slot[i] = (uint32_t)i;
  }
}
__global__ void remap_cols_kernel(const uint32_t* __restrict__ col, uint64_t nnz, const uint32_t* __restrict__ slot,
                                  uint32_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) out[i] = slot[col[i]];
}
template <class T>
__global__ void xh_build_kernel(const T* __restrict__ x, const uint32_t* __restrict__ hot_cols, uint32_t k,
                                T* __restrict__ xh) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) xh[i] = x[hot_cols[i]];
}

// Decide the hot set: the largest threshold t whose columns {cnt >= t} fit the table budget.  Disabled when x
// fits the budget anyway, when cols >= 2^31 (no spare bit), or when the hot set covers < 40 % of the gathers.
// Column reference counts of a matrix and the same counts in descending order: computed once per analysis and shared by
// the hot-column table of the plain path and the slab tables of the sliced one.
struct ColStats {
  DevBuf cnt, sorted, out;
  uint32_t cols = 0;
};
static int col_stats(const spam_csr* a, ColStats* cs, hipStream_t s) {
  cs->cols = (uint32_t)a->cols;
  SPAM_TRY(cs->cnt.alloc((uint64_t)cs->cols * 4));
  SPAM_TRY(cs->sorted.alloc((uint64_t)cs->cols * 4));
  SPAM_TRY(cs->out.alloc(16));
  SPAM_HIP_TRY(hipMemsetAsync(cs->cnt.p, 0, (uint64_t)cs->cols * 4, s));
  if (a->nnz >= (4u << 20) && a->nnz < 0xFFFFFFFFull) {
    DevBuf sc, first;
    SPAM_TRY(sc.alloc(a->nnz * 4));
    SPAM_TRY(first.alloc((uint64_t)cs->cols * 4));
    int bits = 1;
    while (bits < 32 && ((uint64_t)(cs->cols - 1) >> bits)) ++bits;
    SPAM_TRY(sort_u32_asc(a->col_idx, sc.as<uint32_t>(), a->nnz, bits, s));
    col_runs_kernel<<<8192, 256, 0, s>>>(sc.as<uint32_t>(), a->nnz, first.as<uint32_t>(), cs->cnt.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
    col_runs_end_kernel<<<8192, 256, 0, s>>>(sc.as<uint32_t>(), a->nnz, first.as<uint32_t>(), cs->cnt.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
    SPAM_HIP_TRY(hipStreamSynchronize(s));  // sc / first go back to the allocator
  } else {
    col_count_kernel<<<8192, 256, 0, s>>>(a->col_idx, a->nnz, cs->cnt.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
  }
  return sort_u32_desc(cs->cnt.as<uint32_t>(), cs->sorted.as<uint32_t>(), cs->cols, s);
}
// Smallest threshold t >= 1 whose column set {cnt >= t} has at most kmax members (= the (kmax + 1)-th largest count + 1);
// n = their number, mass = their total count.  Two small device reads instead of a 31-step bisection.
static int threshold_from_sorted(const ColStats* cs, uint64_t kmax, hipStream_t s, uint32_t* t_out, uint64_t* n_out,
                                 uint64_t* mass_out) {
  uint32_t t = 1;
  if (kmax < cs->cols) {
    uint32_t v = 0;
    SPAM_HIP_TRY(hipMemcpyAsync(&v, cs->sorted.as<uint32_t>() + kmax, 4, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    t = v + 1u;
  }
  unsigned long long h[2] = {0, 0};
  SPAM_HIP_TRY(hipMemsetAsync(cs->out.p, 0, 16, s));
  const unsigned g = (unsigned)std::min<uint64_t>(((uint64_t)cs->cols + 255) / 256, 4096);
  count_ge_kernel<<<g, 256, 0, s>>>(cs->cnt.as<uint32_t>(), cs->cols, t, cs->out.as<unsigned long long>());
  SPAM_CHECK_LAUNCH();
  SPAM_HIP_TRY(hipMemcpyAsync(h, cs->out.p, 16, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  *t_out = t, *n_out = h[0], *mass_out = h[1];
  return SPAM_OK;
}

template <class T>
static int spmv_hot_analyse(const spam_csr* a, SpmvPlan* p, int hot_kb, hipStream_t s, const ColStats* shared = nullptr) {
  p->hot_kb = hot_kb;
  p->hot_k = 0;
  const uint64_t kmax = (uint64_t)hot_kb * 1024 / sizeof(T);
  if (hot_kb <= 0 || a->cols >= HOT_BIT || a->cols <= 2 * kmax || a->nnz == 0) return SPAM_OK;
  const uint32_t cols = (uint32_t)a->cols;
  ColStats own;
  if (!shared) {
    SPAM_TRY(col_stats(a, &own, s));
    shared = &own;
  }
  const DevBuf& cnt = shared->cnt;
  DevBuf flag, pos, tmp, slot;
  uint32_t lo = 1;
  unsigned long long best[2] = {0, 0};
  {
    uint64_t n = 0, mass = 0;
    SPAM_TRY(threshold_from_sorted(shared, kmax, s, &lo, &n, &mass));
    best[0] = n, best[1] = mass;
  }
  const uint32_t t = lo;
  if (best[0] == 0 || (double)best[1] < 0.40 * (double)a->nnz) return SPAM_OK;
  SPAM_TRY(flag.alloc((uint64_t)cols * 4));
  SPAM_TRY(pos.alloc(((uint64_t)cols + 1) * 8));
  SPAM_TRY(tmp.alloc(scan_tmp_bytes(cols)));
  SPAM_TRY(slot.alloc((uint64_t)cols * 4));
  DevBuf ja, hc, xh;
  SPAM_TRY(ja.alloc(a->nnz * 4));
  SPAM_TRY(hc.alloc(best[0] * 4));
  SPAM_TRY(xh.alloc(best[0] * sizeof(T)));
  const unsigned gc = (unsigned)((cols + 255) / 256);
  hot_flag_kernel<<<gc, 256, 0, s>>>(cnt.as<uint32_t>(), cols, t, flag.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  SPAM_TRY(scan_u32_to_u64(flag.as<uint32_t>(), cols, pos.as<uint64_t>(), tmp.p, s));
  hot_assign_kernel<<<gc, 256, 0, s>>>(flag.as<uint32_t>(), pos.as<uint64_t>(), cols, slot.as<uint32_t>(), hc.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  remap_cols_kernel<<<8192, 256, 0, s>>>(a->col_idx, a->nnz, slot.as<uint32_t>(), ja.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // temporaries go back to the pool at scope exit
  p->hot_k = (uint32_t)best[0];
  p->hot_cover = (double)best[1] / (double)a->nnz;
  p->ja_hot = (uint32_t*)ja.release();
  p->hot_cols = (uint32_t*)hc.release();
  p->xh = xh.release();
  return SPAM_OK;
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}
static SpmvKnobs read_knobs() {
  SpmvKnobs k;
  {
    const int n = env_int("SPAM_SPMV_NPT", 0);
    k.npt = n == 8 || n == 16 ? n : 0;  // 0: by the size of the matrix (build_tile_map)
  }
  k.nt = env_int("SPAM_SPMV_NT", 1);
  k.hot = env_int("SPAM_SPMV_HOT", 2);
  k.hot_kb = k.hot ? env_int("SPAM_SPMV_HOT_KB", 3072) : 0;
  k.sliced = env_int("SPAM_SPMV_SLICED", 1);
  k.slice_kb = env_int("SPAM_SPMV_SLICE_KB", 3072);
  k.lds_kb = env_int("SPAM_SPMV_LDS_KB", -1);
  k.l0_kb = env_int("SPAM_SPMV_L0_KB", 16);
  k.long_min = env_int("SPAM_SPMV_LONG", 0);
  const int t = env_int("SPAM_SPMV_SLICED_TPB", 256), n = env_int("SPAM_SPMV_SLICED_NPT", 8);
  k.tpb = t == 1024 ? 1024 : (t == 512 ? 512 : 256);
  k.snpt = k.tpb == 1024 ? 4 : (k.tpb == 512 ? 8 : (n == 16 ? 16 : 8));
  k.wgs = env_int("SPAM_SPMV_SLICED_WGS", 16);
  k.pipe = env_int("SPAM_SPMV_PIPE", 1);
  k.prof = env_int("SPAM_SPMV_PROF", 0);
  k.rg = env_int("SPAM_SPMV_RG", 1);
  k.rgk.eblk = env_int("SPAM_SPMV_RG_EBLK", k.rgk.eblk);
  k.rgk.nt = env_int("SPAM_SPMV_RG_NT", k.rgk.nt);
  k.rgk.k = env_int("SPAM_SPMV_RG_K", k.rgk.k);
  k.rgk.acc_bits = env_int("SPAM_SPMV_RG_ABITS", 0);
  k.rgk.fold_short = env_int("SPAM_SPMV_RG_SHORT", 1);
  k.rgk.order_min = env_int("SPAM_SPMV_RG_ORDER", k.rgk.order_min);
  k.rgk.fmt = env_int("SPAM_SPMV_RG_FMT", k.rgk.fmt);
  k.rgk.keep_vsrc = env_int("SPAM_SPMV_RG_VSRC", k.rgk.keep_vsrc);
  k.rgk.prof = k.prof;

  return k;
}

// =====================================================================================================
// XCD-sliced SpMV
// =====================================================================================================
constexpr int NSLAB = SPMV_NSLAB;             // one table slab per XCD of MI355X
constexpr int NPART = SPMV_NPART;             // + the cold part: entries whose column is in no table (gathered from x)
constexpr uint32_t NO_DEST = 0xFFFFFFFFu;
constexpr int HEAD_STRIDE = SPMV_HEAD_STRIDE; // u32 words between two queue heads

template <class T>
struct SlabDesc {
  const uint32_t* row_ptr;  // u32[nl + 1]: entry offsets of the compact long rows inside this slab
  const uint32_t* col;      // table slot inside the slab (the column index itself in the cold part)
  const T* val;
  const T* xh;              // this slab's packed x table, S slots (unused in the cold part: x itself)
  T* y;                     // partial sums over the long rows, T[nl]
  const uint32_t* fo;       // first_owned of this slab's tiles
  T* carry;
  uint32_t nnz, ntiles;
};

struct SlicedPlan {
  uint32_t S = 0, L = 0;        // table slots per slab; the first L hold the most referenced columns, contiguous (they
                                // stay in the CU's 32 KB L1; with use_lds every workgroup copies them to LDS instead)
  bool use_lds = false;
  bool pipelined = true;        // software-pipelined persistent kernel (SPAM_SPMV_PIPE=0: the plain loop)
  uint32_t ns = 0;              // short rows (with y_covered: the empty rows too, as rows of zero entries)
  bool y_covered = false;       // every row of y is written by the reduce or a short-row block: no zero-fill of y per product
  uint32_t nl = 0;              // long rows
  uint32_t long_min = 0;        // rows with >= long_min entries are long
  int tpb = 256, npt = 16;
  double cover_table = 0, cover_single = 0;
  uint64_t n_table_cols = 0, nnz_long = 0;
  uint32_t total_tiles = 0;
  unsigned grid = 0;
  uint32_t* dest = nullptr;       // u32[cols]: flat table index of a table column, NO_DEST otherwise
  void* xh = nullptr;             // T[NSLAB * S]
  uint32_t* long_rows = nullptr;  // u32[nl]
  uint32_t* short_rows = nullptr; // u32[ns]
  uint32_t* srow_ptr = nullptr;   // u32[NPART][nl + 1]
  uint32_t* scol = nullptr;
  void* sval = nullptr;
  uint32_t* sfo = nullptr;
  void* scarry = nullptr;
  void* partial = nullptr;        // T[NPART][nl]
  void* descs = nullptr;          // SlabDesc<T>[NPART] on the device
  uint32_t* heads = nullptr;      // tile queue heads, HEAD_STRIDE words apart
  spam_csr* residual = nullptr;   // the non-empty short rows, compacted (row_ptr32 has ns + 1 entries)
  // kept for spam_csr_values_changed (re-gather the values into sval / the residual without a new analysis)
  uint32_t* code = nullptr;       // u32[cols]: table slot of a table column, the column itself otherwise
  uint8_t* slab = nullptr;        // u8[cols]: part of a column (NSLAB = cold)
  uint32_t* res32 = nullptr;      // u32[rows + 1]: entry offsets of the short rows inside the residual
  uint64_t* sbase_d = nullptr;    // u64[NPART]: first entry of every part inside scol / sval
  RgPlan* rg = nullptr;           // the long rows in row-group form (then srow_ptr .. descs above stay null)
  int hot = 2, prof = 0, wgs = 16;  // knobs the launches need
  float ms_plain = 0, ms_sliced = 0;
};

static void sliced_plan_free(SlicedPlan* q) {
  if (!q) return;
  void* ptrs[] = {q->dest, q->xh, q->long_rows, q->short_rows, q->srow_ptr, q->scol, q->sval, q->sfo, q->scarry, q->partial, q->descs, q->heads,
                  q->code, q->slab, q->res32, q->sbase_d};
  for (void* p : ptrs)
    if (p) dev_free(p);
  if (q->rg) rg_plan_free(q->rg);
  if (q->residual) {
    spam_csr* r = q->residual;
    if (r->spmv_plan) spmv_plan_free(r->spmv_plan);
    if (r->row_ptr32) dev_free(r->row_ptr32);
    if (r->col_idx) dev_free(r->col_idx);
    if (r->vals) dev_free(r->vals);
    delete r;
  }
  delete q;
}

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v;
}

// One tile gathers either from a slab table (the first L slots from LDS when the workgroup holds them) or, in the
// cold part, from x itself with non-temporal loads; the choice is uniform per tile, so no lane diverges.
template <class T>
struct GatherSliced {
  const T SPAM_GLB* __restrict__ base;
  const T* lds;
  uint32_t L;
  bool cold;
  __device__ __forceinline__ T operator()(uint32_t c) const {
    if (cold) return __builtin_nontemporal_load(base + c);
    if (L == 0) return base[c];
    // branch-free: a per-lane branch between an LDS read and a global load makes the compiler wait for every gather
    // before it issues the next one (the whole tile then costs 8 serial L2 round trips).  Lanes served by LDS point
    // their global load at slot 0 — identical addresses collapse into one request per wave instruction.
    const bool in_lds = c < L;
    const T g = base[in_lds ? 0u : c];
    const T l = lds[in_lds ? c : 0u];
    return in_lds ? l : g;
  }
};

// The three cases of GatherSliced as types of their own: the pipelined kernel picks one per part and tile, OUTSIDE the
// unrolled gather loop of tile_issue.  With the (uniform) case distinction inside the functor the compiler kept a
// branch per gather, every gather in a basic block of its own, and waited for each (vmcnt(0)) before issuing the next.
template <class T>
struct GatherCold {
  const T SPAM_GLB* __restrict__ base;
  __device__ __forceinline__ T operator()(uint32_t c) const { return __builtin_nontemporal_load(base + c); }
};
template <class T>
struct GatherTable {
  const T SPAM_GLB* __restrict__ base;
  __device__ __forceinline__ T operator()(uint32_t c) const { return base[c]; }
};
template <class T>
struct GatherLevel {  // slots below L from the LDS copy of the table's head, the others from the table (see GatherSliced)
  const T SPAM_GLB* __restrict__ base;
  const T* lds;
  uint32_t L;
  __device__ __forceinline__ T operator()(uint32_t c) const {
    const bool in_lds = c < L;
    const T g = base[in_lds ? 0u : c];
    const T l = lds[in_lds ? c : 0u];
    return in_lds ? l : g;
  }
};

// Persistent workgroups: each learns the XCD it runs on, works through that XCD's slab queue (its gathers then hit
// the slab table in the local L2, the hottest L slots in LDS), and afterwards helps the other slabs finish.
template <class T, int TPB, int NPT, bool LDS, int PROF = 0>
__global__ __launch_bounds__(TPB) void spmv_sliced_kernel(const SlabDesc<T>* __restrict__ descs, const T* __restrict__ x,
                                                         uint32_t L, uint32_t* __restrict__ heads) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  __shared__ TileSmem<T, TPB, NPT> sm;
  __shared__ uint32_t s_t;
  const uint32_t home = xcc_id() & (NSLAB - 1);
  T* tab = reinterpret_cast<T*>(dyn_lds);
  if (LDS) {
    const T* src = descs[home].xh;
    for (uint32_t j = threadIdx.x; j < L; j += TPB) tab[j] = src[j];
  }
  // own slab first, then the cold part (any XCD), then whatever the other XCDs have left
  for (uint32_t d = 0; d < NPART; ++d) {
    const uint32_t k = d == 0 ? home : (d == 1 ? (uint32_t)NSLAB : ((home + d - 1) & (NSLAB - 1)));
    const SlabDesc<T> sd = descs[k];
    const bool cold = k == NSLAB;
    const GatherSliced<T> gx{(const T SPAM_GLB*)(cold ? x : sd.xh), tab, (LDS && d == 0) ? L : 0u, cold};
    uint32_t* head = heads + k * HEAD_STRIDE;
    if (threadIdx.x == 0) s_t = atomicAdd(head, 1u);
    __syncthreads();
    uint32_t t = s_t;
    __syncthreads();
    while (t < sd.ntiles) {
      uint32_t nxt = 0;
      if (threadIdx.x == 0) nxt = atomicAdd(head, 1u);  // the next ticket travels while this tile is processed
      spmv_tile_body<T, uint32_t, TPB, NPT, true, GatherSliced<T>, PROF>(
          sm, (const uint32_t SPAM_GLB*)sd.row_ptr, (const uint32_t SPAM_GLB*)sd.col, (const T SPAM_GLB*)sd.val, gx,
          (T SPAM_GLB*)sd.y, (const uint32_t SPAM_GLB*)sd.fo, (T SPAM_GLB*)sd.carry, sd.nnz, t);
      long long q0 = 0;
      if (PROF) q0 = clock64();
      if (threadIdx.x == 0) s_t = nxt;
      __syncthreads();
      t = s_t;
      __syncthreads();
      if (PROF && threadIdx.x == 0) atomicAdd(&g_spmv_prof[4], (unsigned long long)(clock64() - q0));
    }
  }
}

// The same work, software-pipelined: while a workgroup sums the rows of tile i out of LDS, the loads of its next
// tile (stream, gathers, row_ptr slice) are already in flight, so the L2 keeps receiving gathers from every resident
// workgroup all the time (with the unpipelined loop a workgroup's phase 2 and its three dependent load latencies
// per tile leave the L2 request queues a third empty: 175 instead of 230 G gathers/s).  Tickets are drawn two ahead.
template <class T, int TPB, int NPT, bool LDS, int PROF = 0>
__global__ __launch_bounds__(TPB) void spmv_sliced_pipe_kernel(const SlabDesc<T>* __restrict__ descs, const T* __restrict__ x,
                                                              uint32_t L, uint32_t* __restrict__ heads) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  __shared__ TileSmem<T, TPB, NPT> sm;
  __shared__ uint32_t s_t[2];
  constexpr uint32_t TILE = TPB * NPT;
  const uint32_t home = xcc_id() & (NSLAB - 1);
  T* tab = reinterpret_cast<T*>(dyn_lds);
  if (LDS) {
    const T* src = descs[home].xh;
    for (uint32_t j = threadIdx.x; j < L; j += TPB) tab[j] = src[j];
  }
  TileRegs<T, uint32_t, TPB, NPT> r;
  for (uint32_t d = 0; d < NPART; ++d) {
    const uint32_t k = d == 0 ? home : (d == 1 ? (uint32_t)NSLAB : ((home + d - 1) & (NSLAB - 1)));
    const SlabDesc<T> sd = descs[k];
    const bool cold = k == NSLAB;
    const GatherSliced<T> gx{(const T SPAM_GLB*)(cold ? x : sd.xh), tab, (LDS && d == 0) ? L : 0u, cold};
    const uint32_t SPAM_GLB* row_ptr = (const uint32_t SPAM_GLB*)sd.row_ptr;
    const uint32_t SPAM_GLB* col = (const uint32_t SPAM_GLB*)sd.col;
    const T SPAM_GLB* val = (const T SPAM_GLB*)sd.val;
    const uint32_t SPAM_GLB* fo = (const uint32_t SPAM_GLB*)sd.fo;
    T SPAM_GLB* y = (T SPAM_GLB*)sd.y;
    T SPAM_GLB* carry = (T SPAM_GLB*)sd.carry;
    const uint32_t nfull = sd.nnz / TILE;  // tiles [0, nfull) are full, a last shorter one may follow
    uint32_t* head = heads + k * HEAD_STRIDE;
    if (threadIdx.x == 0) {
      s_t[0] = atomicAdd(head, 1u);
      s_t[1] = atomicAdd(head, 1u);
    }
    __syncthreads();
    uint32_t t_cur = s_t[0], t_next = s_t[1];
    __syncthreads();
    auto issue = [&](uint32_t t) {  // (one straight-line instance of tile_issue per gather kind)
      if (cold)
        tile_issue<T, uint32_t, TPB, NPT, true>(r, row_ptr, col, val, GatherCold<T>{gx.base}, fo, t);
      else if (gx.L == 0)
        tile_issue<T, uint32_t, TPB, NPT, true>(r, row_ptr, col, val, GatherTable<T>{gx.base}, fo, t);
      else
        tile_issue<T, uint32_t, TPB, NPT, true>(r, row_ptr, col, val, GatherLevel<T>{gx.base, gx.lds, gx.L}, fo, t);
    };
    bool loaded = t_cur < nfull;
    if (loaded) issue(t_cur);
    while (t_cur < sd.ntiles) {
      uint32_t t_nn = 0;
      if (threadIdx.x == 0) {
        t_nn = atomicAdd(head, 1u);
        sm.n_long = 0, sm.n_big = 0;
      }
      if (loaded) {
        long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0;
        if (PROF == 1) q0 = clock64();
        tile_finish<T, uint32_t, TPB, NPT>(r, sm, t_cur);
        const uint32_t f0 = r.f0, f1 = r.f1;
        if (PROF == 1) q1 = clock64();
        __syncthreads();
        if (PROF == 1) q2 = clock64();
        loaded = t_next < nfull;
        if (loaded) issue(t_next);
        if (PROF == 1) q3 = clock64();
        const uint64_t base = (uint64_t)t_cur * TILE;
        if (PROF != 2)  // 2: experiment, loads and products only (wrong results)
          tile_phase2<T, uint32_t, TPB, NPT, (PROF == 1 ? 1 : 0)>(sm, row_ptr, y, carry, f0, f1, base, base + TILE, t_cur, &q4);
        if (PROF == 1 && threadIdx.x == 0) {  // [0] wait for gathers + products, [1] barrier, [2] issue next, [3] short rows, [4] long rows
          const long long q5 = clock64();
          atomicAdd(&g_spmv_prof[0], (unsigned long long)(q1 - q0));
          atomicAdd(&g_spmv_prof[1], (unsigned long long)(q2 - q1));
          atomicAdd(&g_spmv_prof[2], (unsigned long long)(q3 - q2));
          atomicAdd(&g_spmv_prof[3], (unsigned long long)(q4 - q3));
          atomicAdd(&g_spmv_prof[4], (unsigned long long)(q5 - q4));
          atomicAdd(&g_spmv_prof[5], 1ull);
        }
      } else {
        spmv_tile_body<T, uint32_t, TPB, NPT, true>(sm, row_ptr, col, val, gx, y, fo, carry, sd.nnz, t_cur);
        loaded = t_next < nfull;
        if (loaded) issue(t_next);
      }
      long long q6 = 0;
      if (PROF == 1) q6 = clock64();
      if (threadIdx.x == 0) s_t[0] = t_nn;
      __syncthreads();
      t_cur = t_next;
      t_next = s_t[0];
      __syncthreads();
      if (PROF == 1 && threadIdx.x == 0) atomicAdd(&g_spmv_prof[6], (unsigned long long)(clock64() - q6));
    }
  }
}

template <class T>
__global__ void sliced_fixup_kernel(const SlabDesc<T>* __restrict__ descs, uint32_t tile) {
  const SlabDesc<T> sd = descs[blockIdx.y];
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 || t >= sd.ntiles) return;
  const uint32_t f = sd.fo[t];
  if (f == 0) return;
  const uint64_t rs = sd.row_ptr[f];
  if (!(rs > t * tile)) return;
  if (sd.fo[t - 1] == f) return;
  T acc = sd.y[f - 1];
  for (uint64_t u = t; u < sd.ntiles && sd.fo[u] == f && rs > u * tile; ++u) acc += sd.carry[u];
  sd.y[f - 1] = acc;
}

// y[long row] = sum of the 8 slab partials and the cold partial, in that order (deterministic)
template <class T>
__global__ void sliced_reduce_kernel(const T* __restrict__ partial, const uint32_t* __restrict__ long_rows, uint32_t nl,
                                     T* __restrict__ y) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nl) return;
  T acc = partial[i];
#pragma unroll
  for (int k = 1; k < NPART; ++k) acc += partial[(uint64_t)k * nl + i];
  y[long_rows[i]] = acc;
}

// The non-empty rows of fewer than long_min entries (compacted): one lane per row, summed left to right — the
// oracle's order, so these rows are bit-identical to it.  The first two entries are loaded together.  The kernel
// moves 0.6 GB for 9 M entries on the bench matrix (a third of its gathers miss the one table these rows share and
// fetch a 128-byte line each): it runs at the fabric rate; two or four rows per lane were measured slower.
template <class T, int HOT>
__global__ __launch_bounds__(256) void spmv_short_rows_kernel(const uint32_t* __restrict__ res_ptr,
                                                              const uint32_t* __restrict__ col, const T* __restrict__ val,
                                                              const T* __restrict__ x, const T* __restrict__ xh,
                                                              const uint32_t* __restrict__ short_rows, uint32_t ns,
                                                              T* __restrict__ y) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const uint32_t s = res_ptr[i], e = res_ptr[i + 1];
  const uint32_t r = short_rows[i];
  if (s >= e) {  // a row of zero entries (listed when the plan covers all of y): nothing to gather
    y[r] = T(0);
    return;
  }
  const uint32_t s1 = s + 1 < e ? s + 1 : s;
  const uint32_t c0 = col[s], c1 = col[s1];
  const T v0 = val[s], v1 = val[s1];
  const T x0 = gather_x<T, HOT>(x, xh, c0), x1 = gather_x<T, HOT>(x, xh, c1);
  T acc = v0 * x0;
  if (s + 1 < e) acc += v1 * x1;
  for (uint32_t p = s + 2; p < e; ++p) acc += val[p] * gather_x<T, HOT>(x, xh, col[p]);
  y[r] = acc;
}

// Rebuild the 8 slab tables from x (column order: x and dest stream, the table writes of one workgroup are
// contiguous runs inside each slab), reset the tile queues and zero y (the empty rows; every other row is written later).
template <class T>
__global__ void sliced_build_tables_kernel(const T* __restrict__ x, const uint32_t* __restrict__ dest, uint32_t cols,
                                           T* __restrict__ xh, uint32_t* __restrict__ heads, T* __restrict__ y,
                                           uint32_t rows, T* __restrict__ cp, const uint64_t* __restrict__ piece_pairs,
                                           uint32_t n_piece) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= NPART) heads[i * HEAD_STRIDE] = 0;
  if (i < n_piece) cp[piece_pairs[i]] = T(0);  // pair sums that pieces of a block add to start every product at zero
  if (i < rows) y[i] = T(0);
  if (i >= cols) return;
  const uint32_t d = dest[i];
  if (d != NO_DEST) xh[d] = x[i];
}

// The same with 16-byte accesses of x and y (V = 16 / sizeof(T) columns / rows per thread); x and y must be 16-byte aligned.
template <class T, int V>
__global__ void sliced_build_tables_vec_kernel(const T* __restrict__ x, const uint32_t* __restrict__ dest, uint32_t cols,
                                               T* __restrict__ xh, uint32_t* __restrict__ heads, T* __restrict__ y,
                                               uint32_t rows, T* __restrict__ cp, const uint64_t* __restrict__ piece_pairs,
                                               uint32_t n_piece) {
  typedef T TV __attribute__((ext_vector_type(V)));
  typedef uint32_t UV __attribute__((ext_vector_type(V)));
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i = j * V;
  if (j <= NPART) heads[j * HEAD_STRIDE] = 0;
  if (j < n_piece) cp[piece_pairs[j]] = T(0);  // pair sums that pieces of a block add to start every product at zero
  if (i + V <= rows) {
    TV z;
#pragma unroll
    for (int v = 0; v < V; ++v) z[v] = T(0);
    *reinterpret_cast<TV*>(y + i) = z;
  } else {
    for (uint64_t r = i; r < rows; ++r) y[r] = T(0);
  }
  if (i + V <= cols) {
    const UV d = *reinterpret_cast<const UV*>(dest + i);
    const TV xv = *reinterpret_cast<const TV*>(x + i);
#pragma unroll
    for (int v = 0; v < V; ++v)
      if (d[v] != NO_DEST) xh[d[v]] = xv[v];
  } else {
    for (uint64_t c = i; c < cols; ++c) {
      const uint32_t d = dest[c];
      if (d != NO_DEST) xh[d] = x[c];
    }
  }
}

// ---- analysis kernels ---------------------------------------------------------------------------------
__global__ void sliced_level_flags_kernel(const uint32_t* __restrict__ cnt, uint32_t cols, uint32_t t_lds, uint32_t t_tab,
                                          uint32_t* __restrict__ f0, uint32_t* __restrict__ f1) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols) return;
  const uint32_t c = cnt[i];
  f0[i] = c >= t_lds;
  f1[i] = c >= t_tab && c < t_lds;
}
__device__ __forceinline__ uint32_t mix32(uint32_t v) {
  v ^= v >> 16, v *= 0x7FEB352Du, v ^= v >> 15, v *= 0x846CA68Bu, v ^= v >> 16;
  return v;
}
// Level-0 columns (LDS level) are dealt over the slabs in column order, then the level-1 columns: each group of 8
// consecutive columns of a level gives one column to every slab, so a slab's slots rise with the column index inside
// each level.  Which member goes to which slab is rotated by a hash of the group number: in R-MAT the popularity of
// a column follows its low bits, and a fixed member -> slab rule hands one slab several times the entries of
// another.  Everything else is cold: part NSLAB, gathered from x itself.
__global__ void sliced_assign_kernel(const uint32_t* __restrict__ f0, const uint32_t* __restrict__ f1,
                                     const uint64_t* __restrict__ p0, const uint64_t* __restrict__ p1, uint32_t cols,
                                     uint32_t L, uint32_t S, uint32_t* __restrict__ code,
                                     uint8_t* __restrict__ slab, uint32_t* __restrict__ dest) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols) return;
  if (f0[i]) {
    const uint64_t p = p0[i];
    const uint32_t j = (uint32_t)(p / NSLAB), k = ((uint32_t)(p % NSLAB) + mix32(j)) & (NSLAB - 1);
    code[i] = j, slab[i] = (uint8_t)k, dest[i] = k * S + j;
  } else if (f1[i]) {
    const uint64_t p = p1[i];
    const uint32_t g = (uint32_t)(p / NSLAB), j = L + g, k = ((uint32_t)(p % NSLAB) + mix32(g ^ 0x9E3779B9u)) & (NSLAB - 1);
    code[i] = j, slab[i] = (uint8_t)k, dest[i] = k * S + j;
  } else {
    code[i] = (uint32_t)i, slab[i] = (uint8_t)NSLAB, dest[i] = NO_DEST;
  }
}
// include_empty: rows without entries join the short rows (as rows of zero entries): every row of y is then written by
// the kernel that owns it — the reduce (sliced rows) or a short-row block — and the zero-fill of y is not needed.
// code / slab of every column back from dest (the row-group form keeps only dest between analyses)
__global__ void sliced_code_from_dest_kernel(const uint32_t* __restrict__ dest, uint32_t cols, uint32_t S,
                                             uint32_t* __restrict__ code, uint8_t* __restrict__ slab) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols) return;
  const uint32_t d = dest[i];
  if (d == NO_DEST) {
    code[i] = (uint32_t)i, slab[i] = (uint8_t)NSLAB;
  } else {
    code[i] = d % S, slab[i] = (uint8_t)(d / S);
  }
}
__global__ void sliced_row_flags_kernel(const uint32_t* __restrict__ row_ptr, uint32_t rows, uint32_t long_min,
                                        int include_empty, uint32_t* __restrict__ is_long,
                                        uint32_t* __restrict__ short_len, uint32_t* __restrict__ is_short) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const uint32_t len = row_ptr[r + 1] - row_ptr[r];
  const bool lg = len >= long_min;
  is_long[r] = lg;
  short_len[r] = lg ? 0u : len;
  is_short[r] = !lg && (len > 0 || include_empty);
}
// compact row list + compact row_ptr of the non-empty short rows
__global__ void sliced_short_list_kernel(const uint32_t* __restrict__ is_short, const uint64_t* __restrict__ pos,
                                         const uint32_t* __restrict__ res_ptr_full, uint32_t rows, uint32_t ns,
                                         uint32_t* __restrict__ short_rows, uint32_t* __restrict__ res_ptr) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > rows) return;
  if (r == rows) {
    res_ptr[ns] = res_ptr_full[rows];
    return;
  }
  if (is_short[r]) {
    short_rows[pos[r]] = (uint32_t)r;
    res_ptr[pos[r]] = res_ptr_full[r];
  }
}
__global__ void sliced_long_list_kernel(const uint32_t* __restrict__ is_long, const uint64_t* __restrict__ pos, uint32_t rows,
                                        uint32_t* __restrict__ long_rows) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows && is_long[r]) long_rows[pos[r]] = (uint32_t)r;
}
// One wavefront per long row: entries per part (8 slabs + cold).
__global__ void sliced_count_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                    const uint8_t* __restrict__ slab, const uint32_t* __restrict__ long_rows, uint32_t nl,
                                    uint32_t* __restrict__ cnt8) {
  const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (w >= nl) return;
  const uint32_t r = long_rows[w];
  const uint32_t s = row_ptr[r], e = row_ptr[r + 1];
  uint32_t n = 0;
  for (uint32_t p = s; p < e; p += 64) {
    const uint32_t k = (p + lane < e) ? slab[col[p + lane]] : NPART;
#pragma unroll
    for (uint32_t kk = 0; kk < NPART; ++kk) {
      const uint64_t m = __ballot(k == kk);
      if (lane == kk) n += (uint32_t)__popcll(m);
    }
  }
  if (lane < NPART) cnt8[(uint64_t)lane * nl + w] = n;
}
// One wavefront per long row: move its entries into the slab arrays, storage order kept inside every (row, slab).
// VALS_ONLY: re-gather the values only (spam_csr_values_changed; the structure has not changed).
template <class T, bool VALS_ONLY = false>
__global__ void sliced_fill_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                   const T* __restrict__ val, const uint32_t* __restrict__ code,
                                   const uint8_t* __restrict__ slab, const uint32_t* __restrict__ long_rows, uint32_t nl,
                                   const uint32_t* __restrict__ srow_ptr, const uint64_t* __restrict__ sbase,
                                   uint32_t* __restrict__ scol, T* __restrict__ sval) {
  const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (w >= nl) return;
  const uint32_t r = long_rows[w];
  const uint32_t s = row_ptr[r], e = row_ptr[r + 1];
  uint64_t mybase = 0;  // lane kk < NPART: next free position of part kk for this row
  if (lane < NPART) mybase = sbase[lane] + srow_ptr[(uint64_t)lane * (nl + 1) + w];
  const uint64_t lt = (1ull << lane) - 1ull;
  for (uint32_t p = s; p < e; p += 64) {
    const bool act = p + lane < e;
    uint32_t c = 0, k = NPART;
    T v = T(0);
    if (act) {
      c = col[p + lane];
      v = val[p + lane];
      k = slab[c];
    }
    uint64_t dst = 0;
#pragma unroll
    for (uint32_t kk = 0; kk < NPART; ++kk) {
      const uint64_t m = __ballot(k == kk);
      const uint64_t b = __shfl(mybase, (int)kk, 64);
      if (k == kk) dst = b + (uint64_t)__popcll(m & lt);
      if (lane == kk) mybase += (uint64_t)__popcll(m);
    }
    if (act) {
      if (!VALS_ONLY) scol[dst] = code[c];
      sval[dst] = v;
    }
  }
}
template <class T, bool VALS_ONLY = false>
__global__ void sliced_residual_fill_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                            const T* __restrict__ val, uint32_t long_min,
                                            const uint32_t* __restrict__ res_ptr, uint32_t rows,
                                            uint32_t* __restrict__ out_col, T* __restrict__ out_val) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const uint32_t s = row_ptr[r], e = row_ptr[r + 1];
  if (e - s >= long_min) return;  // a long row: lives in the slabs
  uint32_t o = res_ptr[r];
  for (uint32_t p = s; p < e; ++p, ++o) {
    if (!VALS_ONLY) out_col[o] = col[p];
    out_val[o] = val[p];
  }
}
__global__ void zero_u32_kernel(uint32_t* __restrict__ p, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

template <class T, int TPB, int NPT, bool LDS>
static int sliced_launch_variant(const SlicedPlan* q, const T* x, hipStream_t s, unsigned* grid_out) {
  const size_t dyn = LDS ? (size_t)q->L * sizeof(T) : 0;  // LDS <=> q->use_lds
  const int prof = grid_out ? 0 : q->prof;  // 1: in-kernel phase timers, 2: no phase 2 (experiment)
  auto kern = q->pipelined ? (prof == 2 ? spmv_sliced_pipe_kernel<T, TPB, NPT, LDS, 2>
                                        : (prof == 1 ? spmv_sliced_pipe_kernel<T, TPB, NPT, LDS, 1>
                                                     : spmv_sliced_pipe_kernel<T, TPB, NPT, LDS, 0>))
                           : spmv_sliced_kernel<T, TPB, NPT, LDS, 0>;
  if (prof == 1 && q->pipelined) {
    unsigned long long h[8] = {0};
    SPAM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_prof), h, sizeof(h)));
    kern<<<q->grid, TPB, dyn, s>>>((const SlabDesc<T>*)q->descs, x, q->L, q->heads);
    SPAM_CHECK_LAUNCH();
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    SPAM_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_spmv_prof), sizeof(h)));
    const double n = (double)(h[5] ? h[5] : 1);
    fprintf(stderr, "[spam] pipelined sliced SpMV, cycles per tile (wave 0): wait gathers + products %.0f, barrier %.0f, "
            "issue next tile %.0f, short rows %.0f, barrier + long rows %.0f, ticket barriers %.0f; tiles %llu, grid %u\n",
            h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[6] / n, h[5], q->grid);
    return SPAM_OK;
  }
  if (grid_out) {
    if (dyn > 48 * 1024) SPAM_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    int per_cu = 0, dev = 0, cus = 0;
    SPAM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, TPB, dyn));
    SPAM_HIP_TRY(hipGetDevice(&dev));
    SPAM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (per_cu < 1) return set_error(SPAM_E_HIP, "sliced SpMV kernel does not fit a CU (%zu B of LDS table)", dyn);
    const int cap = q->wgs;
    if (per_cu > cap) per_cu = cap;
    uint64_t g = (uint64_t)per_cu * (uint64_t)cus;
    if (g > q->total_tiles) g = q->total_tiles;
    if (g == 0) g = 1;
    *grid_out = (unsigned)g;
    return SPAM_OK;
  }
  kern<<<q->grid, TPB, dyn, s>>>((const SlabDesc<T>*)q->descs, x, q->L, q->heads);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T>
static int sliced_dispatch(const SlicedPlan* q, const T* x, hipStream_t s, unsigned* grid_out) {
  if (!grid_out && q->tpb == 256 && q->npt == 16 && !q->use_lds && !q->pipelined) {
    const int prof = q->prof;
    if (prof == 2) {  // experiment: phase 1 only (wrong results)
      spmv_sliced_kernel<T, 256, 16, false, 2><<<q->grid, 256, 0, s>>>((const SlabDesc<T>*)q->descs, x, q->L, q->heads);
      SPAM_CHECK_LAUNCH();
      return SPAM_OK;
    }
    if (prof == 1) {
      unsigned long long h[8] = {0};
      SPAM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_prof), h, sizeof(h)));
      spmv_sliced_kernel<T, 256, 16, false, 1><<<q->grid, 256, 0, s>>>((const SlabDesc<T>*)q->descs, x, q->L, q->heads);
      SPAM_CHECK_LAUNCH();
      SPAM_HIP_TRY(hipStreamSynchronize(s));
      SPAM_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_spmv_prof), sizeof(h)));
      fprintf(stderr, "[spam] sliced SpMV phase cycles per tile (wave 0): stream+gather %.0f, barrier %.0f, short rows %.0f, "
              "barrier+long rows %.0f, dequeue %.0f; tiles %llu, grid %u\n", (double)h[0] / h[5], (double)h[1] / h[5],
              (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5], h[5], q->grid);
      return SPAM_OK;
    }
  }
#define SPAM_SLICED_VARIANT(TPB_, NPT_)                                                         \
  if (q->tpb == TPB_ && q->npt == NPT_)                                                         \
    return q->use_lds ? sliced_launch_variant<T, TPB_, NPT_, true>(q, x, s, grid_out)           \
                : sliced_launch_variant<T, TPB_, NPT_, false>(q, x, s, grid_out)
  SPAM_SLICED_VARIANT(256, 16);
  SPAM_SLICED_VARIANT(256, 8);
  SPAM_SLICED_VARIANT(512, 8);
  SPAM_SLICED_VARIANT(1024, 4);
#undef SPAM_SLICED_VARIANT
  return set_error(SPAM_E_INVALID, "no sliced SpMV kernel for %d threads x %d entries", q->tpb, q->npt);
}

// Build the sliced structure.  Returns SPAM_OK with *out = nullptr when the matrix does not qualify.
template <class T>
static int sliced_analyse(const spam_csr* a, const SpmvKnobs& kn, int force, hipStream_t s, SlicedPlan** out,
                          const ColStats* shared = nullptr) {
  *out = nullptr;
  PhaseClock pc(kn.prof == 4, s);
  const int slice_kb = kn.slice_kb;   // table bytes per XCD (of its 4 MB L2)
  // the hottest slots of every slab are laid out contiguously (they stay in the CU's L1) and, when they cover enough
  // of the gathers to pay for the lost occupancy, copied to LDS by every workgroup.  SPAM_SPMV_LDS_KB: -1 decide
  // (16 KB if that level covers >= 15 % of the entries), 0 never, > 0 that many KB.
  const int lds_cfg = kn.lds_kb;
  const int l0_kb = lds_cfg > 0 ? lds_cfg : kn.l0_kb;
  const uint64_t smax = (uint64_t)slice_kb * 1024 / sizeof(T);
  uint64_t lmax = (uint64_t)(l0_kb > 0 ? l0_kb : 0) * 1024 / sizeof(T);
  if (lmax > smax) lmax = smax;
  if (!a->row_ptr32 || a->nnz == 0 || a->cols >= 0x7FFFFFFFull || a->rows >= 0xFFFFFFFFull || smax == 0) return SPAM_OK;
  if (a->cols + NSLAB * (smax + 32) >= 0xFFFFFFFFull) return SPAM_OK;
  // Candidates (the timed comparison of spam_spmv_prepare has the last word).  The tile form gains from the tables only:
  // x larger than one L2, eight tables covering clearly more gathers than one.  The row-group form also wins where x
  // fits an L2 — its gathers are sorted by slot and served by the L1s, its sums stay in LDS — from a few million
  // entries on (R-MAT scale 18 .. 21: 1.5 .. 2.3 x the plain path, tools/size_sweep.py); below that three launches
  // cost what the product does.
  if (!force && a->nnz < (1u << 20)) return SPAM_OK;
  bool tile_form_candidate = a->cols * sizeof(T) > (4u << 20);
  if (!force && !kn.rg && !tile_form_candidate) return SPAM_OK;
  if (!force && kn.rg && a->nnz < (3u << 20)) return SPAM_OK;
  const uint32_t cols = (uint32_t)a->cols, rows = (uint32_t)a->rows;
  const unsigned gc = (unsigned)(((uint64_t)cols + 255) / 256), gr = (unsigned)(((uint64_t)rows + 255) / 256);

  // -- columns: histogram, level thresholds --------------------------------------------------------------
  ColStats own;
  if (!shared) {
    SPAM_TRY(col_stats(a, &own, s));
    shared = &own;
  }
  const DevBuf& cnt = shared->cnt;
  uint32_t t_tab = 1, t_lds = 0x7FFFFFFFu, t_single = 1;
  uint64_t n_tab = 0, m_tab = 0, n_lds = 0, m_lds = 0, n_single = 0, m_single = 0;
  SPAM_TRY(threshold_from_sorted(shared, NSLAB * smax, s, &t_tab, &n_tab, &m_tab));
  SPAM_TRY(threshold_from_sorted(shared, smax, s, &t_single, &n_single, &m_single));
  if (lmax) SPAM_TRY(threshold_from_sorted(shared, NSLAB * lmax, s, &t_lds, &n_lds, &m_lds));
  if (n_lds == 0) t_lds = 0x7FFFFFFFu;
  const double cover_tab = (double)m_tab / (double)a->nnz, cover_single = (double)m_single / (double)a->nnz;
  if (cover_tab - cover_single < 0.10) tile_form_candidate = false;  // eight tables would hold little more than one
  if (!force && !kn.rg && !tile_form_candidate) return SPAM_OK;
  const uint32_t line_slots = 128 / sizeof(T);
  uint32_t L = (uint32_t)((n_lds + NSLAB - 1) / NSLAB);
  L = (L + line_slots - 1) / line_slots * line_slots;  // the level-1 part of a slab table starts on a 128-byte line
  uint32_t S1 = (uint32_t)((n_tab - n_lds + NSLAB - 1) / NSLAB);
  S1 = std::max<uint32_t>(line_slots, (S1 + line_slots - 1) / line_slots * line_slots);
  const uint32_t S = L + S1;

  pc.stamp("  sliced: thresholds");
  SlicedPlan* q = new (std::nothrow) SlicedPlan();
  if (!q) return set_error(SPAM_E_OOM, "host allocation failed");
  struct Guard {
    SlicedPlan* q;
    ~Guard() { if (q) sliced_plan_free(q); }
  } guard{q};
  q->use_lds = L > 0 && (lds_cfg > 0 || (lds_cfg < 0 && (double)m_lds >= 0.15 * (double)a->nnz));
  q->pipelined = kn.pipe != 0;
  q->S = S, q->L = L, q->cover_table = cover_tab, q->cover_single = cover_single, q->n_table_cols = n_tab;
  q->tpb = kn.tpb, q->npt = kn.snpt, q->hot = kn.hot, q->prof = kn.prof, q->wgs = kn.wgs;
  const uint32_t tile = (uint32_t)(q->tpb * q->npt);
  {
    // rows worth 8 partials: a long row costs ~160 B of partial/row_ptr traffic, a short-path entry that misses
    // the single table costs ~15 ps more than an L2 hit -> long when len * miss > ~1.8
    const double miss = std::max(0.05, 1.0 - cover_single);
    int lm = (int)std::ceil(1.8 / miss);
    lm = std::min(64, std::max(2, lm));
    // the row-group form keeps one accumulator per (row, part) pair that exists, so short rows are cheaper sliced
    // (12 B entry + 16 B pair sum + their share of the reduce) than gathered from x past the tables (a 128-byte line
    // per miss) — down to rows of 3 entries; rows of one or two run as unsliced blocks of the same kernel (measured on
    // the R-MAT bench matrix: thresholds 1 / 2 / 3 / 4 / 5 / 7 / 9 / 13 / 17 -> 0.578 / 0.579 / 0.567 / 0.567 / 0.568 / 0.569 /
    // 0.569 / 0.576 / 0.603 ms; reading those rows' table columns from the slab tables instead of x: no difference)
    q->long_min = (uint32_t)(kn.long_min > 0 ? kn.long_min : (kn.rg ? 3 : lm));
  }

  DevBuf f0, f1, p0, p1, tmp, code, slab, dest;
  SPAM_TRY(f0.alloc((uint64_t)cols * 4));
  SPAM_TRY(f1.alloc((uint64_t)cols * 4));
  SPAM_TRY(p0.alloc(((uint64_t)cols + 1) * 8));
  SPAM_TRY(p1.alloc(((uint64_t)cols + 1) * 8));
  SPAM_TRY(tmp.alloc(scan_tmp_bytes(std::max<uint64_t>(cols, rows))));
  SPAM_TRY(code.alloc((uint64_t)cols * 4));
  SPAM_TRY(slab.alloc((uint64_t)cols));
  SPAM_TRY(dest.alloc((uint64_t)cols * 4));
  sliced_level_flags_kernel<<<gc, 256, 0, s>>>(cnt.as<uint32_t>(), cols, t_lds, t_tab, f0.as<uint32_t>(), f1.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  SPAM_TRY(scan_u32_to_u64(f0.as<uint32_t>(), cols, p0.as<uint64_t>(), tmp.p, s));
  SPAM_TRY(scan_u32_to_u64(f1.as<uint32_t>(), cols, p1.as<uint64_t>(), tmp.p, s));
  sliced_assign_kernel<<<gc, 256, 0, s>>>(f0.as<uint32_t>(), f1.as<uint32_t>(), p0.as<uint64_t>(), p1.as<uint64_t>(), cols, L, S,
                                          code.as<uint32_t>(), slab.as<uint8_t>(), dest.as<uint32_t>());
  SPAM_CHECK_LAUNCH();

  // -- rows: long / short ----------------------------------------------------------------------------------
  DevBuf is_long, short_len, is_short, poss, posl, res64, res32;
  SPAM_TRY(is_long.alloc((uint64_t)rows * 4));
  SPAM_TRY(short_len.alloc((uint64_t)rows * 4));
  SPAM_TRY(is_short.alloc((uint64_t)rows * 4));
  SPAM_TRY(poss.alloc(((uint64_t)rows + 1) * 8));
  SPAM_TRY(posl.alloc(((uint64_t)rows + 1) * 8));
  SPAM_TRY(res64.alloc(((uint64_t)rows + 1) * 8));
  SPAM_TRY(res32.alloc(((uint64_t)rows + 1) * 4));
  // (round 4 measured: listing the empty rows as short rows of zero entries, so that every row of y has a writer and the
  // zero-fill of y can go, changes nothing on the bench matrix — 0.563 against 0.561 ms: the 80 MB of zeros it saves
  // come back as row lists and scattered stores — so the zero-fill stays)
  const bool include_empty = false;
  sliced_row_flags_kernel<<<gr, 256, 0, s>>>(a->row_ptr32, rows, q->long_min, include_empty ? 1 : 0, is_long.as<uint32_t>(),
                                             short_len.as<uint32_t>(), is_short.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  SPAM_TRY(scan_u32_to_u64(is_long.as<uint32_t>(), rows, posl.as<uint64_t>(), tmp.p, s));
  SPAM_TRY(scan_u32_to_u64(is_short.as<uint32_t>(), rows, poss.as<uint64_t>(), tmp.p, s));
  SPAM_TRY(scan_u32_to_u64(short_len.as<uint32_t>(), rows, res64.as<uint64_t>(), tmp.p, s));
  SPAM_TRY(narrow_row_ptr(res64.as<uint64_t>(), (uint64_t)rows + 1, res32.as<uint32_t>(), s));
  uint64_t nl64 = 0, nnz_res = 0, ns64 = 0;
  SPAM_HIP_TRY(hipMemcpyAsync(&nl64, posl.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(&ns64, poss.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(&nnz_res, res64.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (nl64 == 0) return SPAM_OK;  // nothing to slice
  const uint32_t nl = (uint32_t)nl64;
  q->nl = nl;
  q->ns = (uint32_t)ns64;
  q->nnz_long = a->nnz - nnz_res;
  DevBuf long_rows;
  SPAM_TRY(long_rows.alloc((uint64_t)nl * 4));
  sliced_long_list_kernel<<<gr, 256, 0, s>>>(is_long.as<uint32_t>(), posl.as<uint64_t>(), rows, long_rows.as<uint32_t>());
  SPAM_CHECK_LAUNCH();

  // -- residual CSR: the non-empty short rows, compacted ---------------------------------------------------------
  spam_csr* res = new (std::nothrow) spam_csr();
  if (!res) return set_error(SPAM_E_OOM, "host allocation failed");
  q->residual = res;
  res->rows = q->ns, res->cols = a->cols, res->nnz = nnz_res, res->dtype = a->dtype, res->is_sorted = a->is_sorted;
  DevBuf short_rows;
  {
    DevBuf rc, rv, rp;
    SPAM_TRY(rc.alloc((nnz_res + 4) * 4));
    SPAM_TRY(rv.alloc((nnz_res + 4) * sizeof(T)));
    SPAM_TRY(rp.alloc(((uint64_t)q->ns + 1) * 4));
    SPAM_TRY(short_rows.alloc(((uint64_t)q->ns + 1) * 4));
    sliced_residual_fill_kernel<T><<<gr, 256, 0, s>>>(a->row_ptr32, a->col_idx, (const T*)a->vals, q->long_min,
                                                      res32.as<uint32_t>(), rows, rc.as<uint32_t>(), rv.as<T>());
    SPAM_CHECK_LAUNCH();
    sliced_short_list_kernel<<<(unsigned)(((uint64_t)rows + 1 + 255) / 256), 256, 0, s>>>(
        is_short.as<uint32_t>(), poss.as<uint64_t>(), res32.as<uint32_t>(), rows, q->ns, short_rows.as<uint32_t>(),
        rp.as<uint32_t>());
    SPAM_CHECK_LAUNCH();
    res->row_ptr32 = (uint32_t*)rp.release();
    res->col_idx = (uint32_t*)rc.release();
    res->vals = rv.release();
  }
  pc.stamp("  sliced: column maps, row lists, residual");
  // -- the long rows: row-group form (spam_spmv_rg.hip) or slab CSRs ------------------------------------------
  if (kn.rg)
    SPAM_TRY(rg_build<T>(a, slab.as<uint8_t>(), code.as<uint32_t>(), long_rows.as<uint32_t>(), nl, S, res, q->ns, kn.rgk, s, &q->rg));
  q->y_covered = include_empty && q->rg && q->rg->ngs > 0;  // (the short-row blocks of the row-group kernel hold every row that is not sliced)
  pc.stamp("  sliced: row-group plan (rg_build)");
  if (!q->rg && !force && !tile_form_candidate) return SPAM_OK;  // (row groups declined; the guard frees q)
  DevBuf cnt8, sc64, srow, sbase_d;
  DevBuf scol, sval, sfo, scarry, partial, xh, descs, heads;
  if (!q->rg) SPAM_TRY(partial.alloc((uint64_t)NPART * nl * sizeof(T)));  // (the row-group form keeps its pair sums)
  SPAM_TRY(xh.alloc((uint64_t)NSLAB * S * sizeof(T)));
  SPAM_TRY(heads.alloc((NPART + 1) * HEAD_STRIDE * 4));  // + the short-row queue of the row-group form
  SPAM_HIP_TRY(hipMemsetAsync(xh.p, 0, (uint64_t)NSLAB * S * sizeof(T), s));  // unused slots stay defined
  if (!q->rg) SPAM_HIP_TRY(hipMemsetAsync(partial.p, 0, (uint64_t)NPART * nl * sizeof(T), s));  // a part without entries never writes
  if (!q->rg) {
  SPAM_TRY(cnt8.alloc((uint64_t)NPART * nl * 4));
  SPAM_TRY(sc64.alloc(((uint64_t)nl + 1) * 8));
  SPAM_TRY(srow.alloc((uint64_t)NPART * (nl + 1) * 4));
  SPAM_TRY(sbase_d.alloc(NPART * 8));
  const unsigned gw = (unsigned)(((uint64_t)nl * 64 + 255) / 256);
  sliced_count_kernel<<<gw, 256, 0, s>>>(a->row_ptr32, a->col_idx, slab.as<uint8_t>(), long_rows.as<uint32_t>(), nl,
                                         cnt8.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  DevBuf tmp2;
  SPAM_TRY(tmp2.alloc(scan_tmp_bytes(nl)));
  uint64_t ne[NPART];
  for (int k = 0; k < NPART; ++k) {
    SPAM_TRY(scan_u32_to_u64(cnt8.as<uint32_t>() + (uint64_t)k * nl, nl, sc64.as<uint64_t>(), tmp2.p, s));
    SPAM_TRY(narrow_row_ptr(sc64.as<uint64_t>(), (uint64_t)nl + 1, srow.as<uint32_t>() + (uint64_t)k * (nl + 1), s));
    SPAM_HIP_TRY(hipMemcpyAsync(&ne[k], sc64.as<uint64_t>() + nl, 8, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
  }
  uint64_t sbase[NPART + 1];
  uint32_t tile_base[NPART + 1];
  sbase[0] = 0, tile_base[0] = 0;
  for (int k = 0; k < NPART; ++k) {
    sbase[k + 1] = (sbase[k] + ne[k] + 3) & ~3ull;  // 16-byte vector loads start aligned in every slab
    tile_base[k + 1] = tile_base[k] + (uint32_t)((ne[k] + tile - 1) / tile);
  }
  q->total_tiles = tile_base[NPART];
  SPAM_HIP_TRY(hipMemcpyAsync(sbase_d.p, sbase, NPART * 8, hipMemcpyHostToDevice, s));
  SPAM_TRY(scol.alloc((sbase[NPART] + 4) * 4));
  SPAM_TRY(sval.alloc((sbase[NPART] + 4) * sizeof(T)));
  SPAM_TRY(sfo.alloc(((uint64_t)q->total_tiles + NPART) * 4));
  SPAM_TRY(scarry.alloc(((uint64_t)q->total_tiles + 1) * sizeof(T)));
  SPAM_TRY(descs.alloc(NPART * sizeof(SlabDesc<T>)));
  sliced_fill_kernel<T><<<gw, 256, 0, s>>>(a->row_ptr32, a->col_idx, (const T*)a->vals, code.as<uint32_t>(), slab.as<uint8_t>(),
                                           long_rows.as<uint32_t>(), nl, srow.as<uint32_t>(), sbase_d.as<uint64_t>(),
                                           scol.as<uint32_t>(), sval.as<T>());
  SPAM_CHECK_LAUNCH();
  SlabDesc<T> hd[NPART];
  for (int k = 0; k < NPART; ++k) {
    const uint32_t nt = tile_base[k + 1] - tile_base[k];
    uint32_t* fo_k = sfo.as<uint32_t>() + tile_base[k] + k;  // ntiles + 1 entries each
    hd[k].row_ptr = srow.as<uint32_t>() + (uint64_t)k * (nl + 1);
    hd[k].col = scol.as<uint32_t>() + sbase[k];
    hd[k].val = sval.as<T>() + sbase[k];
    hd[k].xh = k < NSLAB ? xh.as<T>() + (uint64_t)k * S : nullptr;
    hd[k].y = partial.as<T>() + (uint64_t)k * nl;
    hd[k].fo = fo_k;
    hd[k].carry = scarry.as<T>() + tile_base[k];
    hd[k].nnz = (uint32_t)ne[k];
    hd[k].ntiles = nt;
    spmv_first_owned_kernel<uint32_t><<<(nt + 1 + 255) / 256, 256, 0, s>>>(hd[k].row_ptr, nl, tile, nt, fo_k);
    SPAM_CHECK_LAUNCH();
  }
  SPAM_HIP_TRY(hipMemcpyAsync(descs.p, hd, sizeof(hd), hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // hd, sbase go out of scope
  }  // !q->rg

  SPAM_HIP_TRY(hipStreamSynchronize(s));  // host-side descriptor arrays and temporaries go out of scope
  q->dest = (uint32_t*)dest.release();
  q->xh = xh.release();
  q->long_rows = (uint32_t*)long_rows.release();
  q->short_rows = (uint32_t*)short_rows.release();
  q->srow_ptr = (uint32_t*)srow.release();
  q->scol = (uint32_t*)scol.release();
  q->sval = sval.release();
  q->sfo = (uint32_t*)sfo.release();
  q->scarry = scarry.release();
  q->partial = partial.release();
  q->descs = descs.release();
  q->heads = (uint32_t*)heads.release();
  if (!q->rg) {  // (the tile form reads both again in spam_csr_values_changed; the row-group form keeps its own source map)
    q->code = (uint32_t*)code.release();
    q->slab = (uint8_t*)slab.release();
  }
  q->res32 = (uint32_t*)res32.release();
  q->sbase_d = (uint64_t*)sbase_d.release();
  if (!q->rg) SPAM_TRY(sliced_dispatch<T>(q, nullptr, s, &q->grid));
  if (!(q->rg && q->rg->ngs)) {  // (in row-group form the short rows are blocks of the same kernel)
    // the short rows' own hot-column table (analysed on the residual's histogram)
    res->spmv_plan = new (std::nothrow) SpmvPlan();  // only the hot-column part is used
    if (!res->spmv_plan) return set_error(SPAM_E_OOM, "host allocation failed");
    SPAM_TRY(spmv_hot_analyse<T>(res, res->spmv_plan, kn.hot_kb, s));
  }
  guard.q = nullptr;
  *out = q;
  return SPAM_OK;
}

// spam_csr_values_changed for the sliced structure: gather the matrix's current values again (same traversal as the
// analysis, values only).
template <class T>
static int sliced_refresh_values(const spam_csr* a, SlicedPlan* q, hipStream_t s) {
  const uint32_t rows = (uint32_t)a->rows;
  const unsigned gw = (unsigned)(((uint64_t)q->nl * 64 + 255) / 256), gr = (unsigned)(((uint64_t)rows + 255) / 256);
  if (q->rg) {
    sliced_residual_fill_kernel<T, true><<<gr, 256, 0, s>>>(a->row_ptr32, a->col_idx, (const T*)a->vals, q->long_min, q->res32, rows,
                                                            q->residual->col_idx, (T*)q->residual->vals);
    SPAM_CHECK_LAUNCH();
    if (!q->rg->vsrc) {
      // The plan was built without the value-source map (4 bytes per padded entry: 0.69 GB for the bench matrix, and most
      // matrices never change their values).  First change: build the row-group plan again, with the map — same
      // inputs, same layout, and its stream already holds the new values; later changes take the short way below.
      RgKnobs kn = a->spmv_plan ? a->spmv_plan->knobs.rgk : RgKnobs();
      kn.keep_vsrc = 1, kn.prof = 0;
      SPAM_HIP_TRY(hipStreamSynchronize(s));  // products queued on the old plan
      RgPlan* fresh = nullptr;
      DevBuf code, slab;  // (not kept by a row-group plan: both follow from dest)
      const uint32_t cols = (uint32_t)a->cols;
      SPAM_TRY(code.alloc((uint64_t)cols * 4));
      SPAM_TRY(slab.alloc((uint64_t)cols));
      sliced_code_from_dest_kernel<<<(unsigned)(((uint64_t)cols + 255) / 256), 256, 0, s>>>(q->dest, cols, q->S, code.as<uint32_t>(),
                                                                                             slab.as<uint8_t>());
      SPAM_CHECK_LAUNCH();
      SPAM_TRY(rg_build<T>(a, slab.as<uint8_t>(), code.as<uint32_t>(), q->long_rows, q->nl, q->S, q->residual, q->ns, kn, s, &fresh));
      SPAM_HIP_TRY(hipStreamSynchronize(s));
      if (!fresh || fresh->npairs != q->rg->npairs || fresh->ntiles != q->rg->ntiles) {
        if (fresh) rg_plan_free(fresh);
        return set_error(SPAM_E_INVALID, "rebuilding the row-group plan for spam_csr_values_changed gave another layout");
      }
      rg_plan_free(q->rg);
      q->rg = fresh;
      return SPAM_OK;
    }
    return rg_refresh_values<T>(a, q->residual, q->rg, s);  // (reads the residual's refreshed values)
  } else if (q->nl) {
    sliced_fill_kernel<T, true><<<gw, 256, 0, s>>>(a->row_ptr32, a->col_idx, (const T*)a->vals, q->code, q->slab, q->long_rows,
                                                   q->nl, q->srow_ptr, q->sbase_d, q->scol, (T*)q->sval);
    SPAM_CHECK_LAUNCH();
  }
  sliced_residual_fill_kernel<T, true><<<gr, 256, 0, s>>>(a->row_ptr32, a->col_idx, (const T*)a->vals, q->long_min, q->res32, rows,
                                                          q->residual->col_idx, (T*)q->residual->vals);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

// Short rows: one lane per row, with the hot-column table the analysis built on the residual's histogram.
template <class T>
static int residual_run(const spam_csr* a, const SlicedPlan* q, const T* x, T* y, hipStream_t s) {
  const spam_csr* res = q->residual;
  const SpmvPlan* pm = res->spmv_plan;
  if (q->ns == 0) return SPAM_OK;
  const unsigned g = (q->ns + 255) / 256;
  if (pm && pm->hot_k && q->hot) {
    xh_build_kernel<T><<<(pm->hot_k + 255) / 256, 256, 0, s>>>(x, pm->hot_cols, pm->hot_k, (T*)pm->xh);
    SPAM_CHECK_LAUNCH();
    spmv_short_rows_kernel<T, 2><<<g, 256, 0, s>>>(res->row_ptr32, pm->ja_hot, (const T*)res->vals, x, (const T*)pm->xh,
                                                   q->short_rows, q->ns, y);
  } else {
    spmv_short_rows_kernel<T, 0><<<g, 256, 0, s>>>(res->row_ptr32, res->col_idx, (const T*)res->vals, x, nullptr,
                                                   q->short_rows, q->ns, y);
  }
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T>
static int sliced_run(const spam_csr* a, const SlicedPlan* q, const T* x, T* y, hipStream_t s) {
  const uint32_t cols = (uint32_t)a->cols;
  const uint32_t rows = (uint32_t)a->rows;
  constexpr int V = 16 / (int)sizeof(T);
  const uint32_t zrows = q->y_covered ? 0u : rows;  // rows of y to zero here: none when every row has an owner that writes it
  T* cp = q->rg ? (T*)q->rg->cp : nullptr;
  const uint64_t* pp = q->rg ? q->rg->piece_pairs : nullptr;
  const uint32_t npp = q->rg ? q->rg->n_piece_pairs : 0u;
  if ((((uintptr_t)x | (uintptr_t)y) & 15u) == 0) {
    const uint64_t nthr = std::max<uint64_t>(((uint64_t)std::max(cols, zrows) + V - 1) / V, npp);
    sliced_build_tables_vec_kernel<T, V><<<(unsigned)((std::max<uint64_t>(nthr, NPART + 1) + 255) / 256), 256, 0, s>>>(
        x, q->dest, cols, (T*)q->xh, q->heads, y, zrows, cp, pp, npp);
  } else {
    const uint64_t nthr = std::max<uint64_t>(std::max<uint64_t>(std::max(cols, zrows), npp), NPART + 1);
    sliced_build_tables_kernel<T><<<(unsigned)((nthr + 255) / 256), 256, 0, s>>>(x, q->dest, cols, (T*)q->xh, q->heads, y, zrows, cp, pp, npp);
  }
  SPAM_CHECK_LAUNCH();
  // (running the short rows beside the slab kernel on a second stream was measured 3 % slower than one after the other)
  if (!(q->rg && q->rg->ngs)) SPAM_TRY(residual_run<T>(a, q, x, y, s));
  if (q->rg) {
    return rg_run<T>(q->rg, (const T*)q->xh, q->S, x, q->heads, y, q->short_rows, s);  // (its own reduce inside)
  } else {
    SPAM_TRY(sliced_dispatch<T>(q, x, s, nullptr));
    const uint32_t tile = (uint32_t)(q->tpb * q->npt);
    uint32_t max_tiles = (q->total_tiles + NPART - 1);  // >= any part's tile count
    sliced_fixup_kernel<T><<<dim3((max_tiles + 255) / 256, NPART), 256, 0, s>>>((const SlabDesc<T>*)q->descs, tile);
    SPAM_CHECK_LAUNCH();
  }

  sliced_reduce_kernel<T><<<(q->nl + 255) / 256, 256, 0, s>>>((const T*)q->partial, q->long_rows, q->nl, y);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

// ---- the plain path ---------------------------------------------------------------------------------------
template <class T, class OffT>
static int plain_run(const spam_csr* a, const OffT* row_ptr, const T* x, T* y, hipStream_t s) {
  const SpmvPlan* p = a->spmv_plan;
  const int nt_cfg = p->knobs.nt;     // non-temporal ja/a stream: keeps L2 for x (+4..6 % on random gathers)
  const int hot_cfg = p->knobs.hot;   // 0 off, 1 per-lane base select, 2 cold gathers non-temporal
  const int npt = p->npt;
  const uint32_t tile = SPMV_TPB * npt;
  const unsigned grid = (unsigned)p->ntiles;
  const int hot = p->hot_k && p->ja_hot ? hot_cfg : 0;
  const uint32_t* col = hot ? p->ja_hot : a->col_idx;
  const T* val = (const T*)a->vals;
  const T* xh = (const T*)p->xh;
  T* carry = (T*)p->carry;
  if (hot) {
    xh_build_kernel<T><<<(p->hot_k + 255) / 256, 256, 0, s>>>(x, p->hot_cols, p->hot_k, (T*)p->xh);
    SPAM_CHECK_LAUNCH();
  }
#define SPAM_LAUNCH_TILE(NPT_, NT_, HOT_)                                                                          \
  spmv_tile_kernel<T, OffT, NPT_, NT_, HOT_><<<grid, SPMV_TPB, 0, s>>>(row_ptr, col, val, x, xh, y, p->first_owned, \
                                                                        carry, a->nnz)
#define SPAM_LAUNCH_HOT(NPT_, NT_)             \
  do {                                         \
    if (hot == 0) SPAM_LAUNCH_TILE(NPT_, NT_, 0); \
    else if (hot == 1) SPAM_LAUNCH_TILE(NPT_, NT_, 1); \
    else SPAM_LAUNCH_TILE(NPT_, NT_, 2);       \
  } while (0)
  if (npt == 8) {
    if (nt_cfg) SPAM_LAUNCH_HOT(8, true); else SPAM_LAUNCH_HOT(8, false);
  } else {
    if (nt_cfg) SPAM_LAUNCH_HOT(16, true); else SPAM_LAUNCH_HOT(16, false);
  }
#undef SPAM_LAUNCH_HOT
#undef SPAM_LAUNCH_TILE
  SPAM_CHECK_LAUNCH();
  if (p->ntiles > 1) {
    spmv_fixup_kernel<T, OffT><<<(unsigned)((p->ntiles + 255) / 256), 256, 0, s>>>(row_ptr, p->first_owned, carry, y,
                                                                                   p->ntiles, tile);
    SPAM_CHECK_LAUNCH();
  }
  return SPAM_OK;
}

template <class F>
static int time_runs(F&& run, hipStream_t s, float* ms_out) {
  hipEvent_t e0, e1;
  SPAM_HIP_TRY(hipEventCreate(&e0));
  SPAM_HIP_TRY(hipEventCreate(&e1));
  int rc = run();  // warm-up (builds lazily created pieces)
  if (rc == SPAM_OK) {
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < 3 && rc == SPAM_OK; ++i) rc = run();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / 3;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// The row -> tile map of the plain path: depends on row_ptr only, one small kernel, no host synchronisation.
// Caller holds the plan's lock (or is the only user of the handle).
template <class OffT>
static int build_tile_map(const spam_csr* a, const OffT* row_ptr, SpmvPlan* p, size_t vsize, hipStream_t s) {
  // tiles of 2048 entries give small matrices more workgroups (R-MAT, 0.2 M .. 65 M entries: 13 -> 10, 18 -> 15, 43 -> 36,
  // 132 -> 124, 536 -> 485 us); from ~100 M entries on 4096 are a little ahead (1257 against 1298 us at 131 M)
  p->npt = p->knobs.npt ? p->knobs.npt : (a->nnz < 100000000ull ? 8 : 16);
  const uint32_t tile = SPMV_TPB * (uint32_t)p->npt;
  p->ntiles = (a->nnz + tile - 1) / tile;
  if (a->nnz == 0) return SPAM_OK;
  DevBuf fo, ca;
  SPAM_TRY(fo.alloc((p->ntiles + 1) * 4));
  SPAM_TRY(ca.alloc(p->ntiles * vsize));
  spmv_first_owned_kernel<OffT><<<(unsigned)((p->ntiles + 1 + 255) / 256), 256, 0, s>>>(row_ptr, (uint32_t)a->rows, tile,
                                                                                       p->ntiles, (uint32_t*)fo.p);
  SPAM_CHECK_LAUNCH();
  p->first_owned = (uint32_t*)fo.release();
  p->carry = ca.release();
  return SPAM_OK;
}

// First use of a handle that was never prepared: the plain tile path without any analysis (no hot-column table, no
// XCD-sliced structure) — asynchronous apart from two small allocations.  spam_spmv_prepare upgrades the plan.
template <class T, class OffT>
static int ensure_plan(const spam_csr* a, const OffT* row_ptr, hipStream_t s) {
  spam_csr* am = const_cast<spam_csr*>(a);
  std::lock_guard<std::mutex> lk(am->plan_mu);
  if (am->spmv_plan) return SPAM_OK;
  SpmvPlan* p = new (std::nothrow) SpmvPlan();
  if (!p) return set_error(SPAM_E_OOM, "host allocation failed");
  p->knobs = read_knobs();
  p->hot_kb = 0;
  if (int e = build_tile_map<OffT>(a, row_ptr, p, sizeof(T), s)) {
    spmv_plan_free(p);
    return e;
  }
  __atomic_store_n(&am->spmv_plan, p, __ATOMIC_RELEASE);
  return SPAM_OK;
}

template <class T, class OffT>
static int spmv_typed(const spam_csr* a, const OffT* row_ptr, const T* x, T* y, hipStream_t s) {
  if (a->nnz == 0) {
    fill_zero_kernel<T><<<(unsigned)((a->rows + 255) / 256), 256, 0, s>>>(y, a->rows);
    SPAM_CHECK_LAUNCH();
    return SPAM_OK;
  }
  if (!__atomic_load_n(&a->spmv_plan, __ATOMIC_ACQUIRE)) SPAM_TRY((ensure_plan<T, OffT>(a, row_ptr, s)));
  const SpmvPlan* pm = a->spmv_plan;
  // one SpMV = several launches over the plan's scratch buffers: enqueue them as one unit (all calls share ONE stream)
  std::lock_guard<std::mutex> run(const_cast<spam_csr*>(a)->run_mu);
  if (pm->sliced) return sliced_run<T>(a, pm->sliced, x, y, s);
  return plain_run<T, OffT>(a, row_ptr, x, y, s);
}

// spam_spmv_prepare: the analysis step of CSR-adaptive, done once per matrix and never inside spam_spmv_dev.
//   * environment knobs are read here, once;
//   * hot-column table of the plain path (one column histogram + a sort of the counts: host-synchronised);
//   * the XCD-sliced structure where the matrix qualifies (SPAM_SPMV_SLICED = 1: kept only if three timed runs on `x`
//     beat the plain path by more than 3 %; 2: always; 0: never).  x = NULL times on a vector of ones.
// The choice fixes the path of every later SpMV on this handle.  Summation order: the plain path and the tile form
// (SPAM_SPMV_RG=0) are deterministic per plan; the row-group form (the default for sliced plans) adds f32 / f64 products
// with LDS atomics, so its results are reproducible to rounding, not bit for bit, from run to run (i64: exact).
template <class T, class OffT>
static int prepare_typed(spam_csr* a, const OffT* row_ptr, const T* x, hipStream_t s) {
  std::lock_guard<std::mutex> lk(a->plan_mu);
  if (a->spmv_plan) {
    SpmvPlan* old = a->spmv_plan;
    a->spmv_plan = nullptr;
    SPAM_HIP_TRY(hipStreamSynchronize(s));  // SpMVs queued on the old plan
    spmv_plan_free(old);
  }
  if (a->nnz == 0) return SPAM_OK;
  SpmvPlan* p = new (std::nothrow) SpmvPlan();
  if (!p) return set_error(SPAM_E_OOM, "host allocation failed");
  struct Guard {
    SpmvPlan* p;
    ~Guard() { if (p) spmv_plan_free(p); }
  } guard{p};
  p->knobs = read_knobs();
  PhaseClock pc(p->knobs.prof == 4, s);
  SPAM_TRY(build_tile_map<OffT>(a, row_ptr, p, sizeof(T), s));
  pc.stamp("row -> tile map");
  ColStats cs;
  // (both analyses give up on 2^31 columns and more, and neither needs the counts of a matrix without entries)
  // ... nor where neither analysis can use them: the hot-column table needs more columns than twice its slots, the sliced
  // candidates start at 3 M entries (row groups) / an x beyond one L2 (tile form) unless forced
  const uint64_t hot_kmax = (uint64_t)(p->knobs.hot_kb > 0 ? p->knobs.hot_kb : 0) * 1024 / sizeof(T);
  const bool hot_eligible = p->knobs.hot_kb > 0 && a->cols > 2 * hot_kmax;
  const bool sliced_eligible = p->knobs.sliced == 2 ||
                               (p->knobs.sliced == 1 && a->row_ptr32 && a->nnz >= (1u << 20) &&
                                (p->knobs.rg ? a->nnz >= (3u << 20) : a->cols * sizeof(T) > (4u << 20)));
  const bool want_stats = a->cols < 0x7FFFFFFFull && a->nnz > 0 && (hot_eligible || sliced_eligible);
  if (want_stats) SPAM_TRY(col_stats(a, &cs, s));
  pc.stamp("column histogram + sort");
  SPAM_TRY(spmv_hot_analyse<T>(a, p, p->knobs.hot_kb, s, want_stats ? &cs : nullptr));
  pc.stamp("plain path: hot-column table");
  const int sl_cfg = p->knobs.sliced;
  if (sl_cfg) {
    SlicedPlan* q = nullptr;
    DevBuf xs, ys;
    const T* xt = x;
    float ms_plain = 0.f;
    bool plain_is_fast = false;
    if (sl_cfg == 1 && sliced_eligible) {
      // The plain path first: a matrix it already streams at 4.5 TB/s of algorithmic bytes (banded ones: 5.3) has
      // nothing to gain from tables or LDS accumulators, and building the sliced structure only to discard it would
      // cost the caller tens to hundreds of milliseconds.
      SPAM_TRY(ys.alloc(a->rows * sizeof(T)));
      if (!xt) {
        SPAM_TRY(xs.alloc(a->cols * sizeof(T)));
        fill_const_kernel<T><<<(unsigned)((a->cols + 255) / 256), 256, 0, s>>>((T*)xs.p, a->cols, T(1));
        SPAM_CHECK_LAUNCH();
        xt = (const T*)xs.p;
      }
      a->spmv_plan = p;  // plain_run reads the plan through the handle
      const int rc = time_runs([&] { return plain_run<T, OffT>(a, row_ptr, xt, (T*)ys.p, s); }, s, &ms_plain);
      a->spmv_plan = nullptr;
      SPAM_TRY(rc);
      p->ms_plain = ms_plain;
      const double bytes = (double)a->nnz * (4.0 + sizeof(T)) + ((double)a->rows + 1) * 4.0 + ((double)a->cols + (double)a->rows) * sizeof(T);
      plain_is_fast = ms_plain > 0.f && bytes / ((double)ms_plain * 1e-3) >= 4.5e12;
      pc.stamp("plain path timed");
    }
    if (!plain_is_fast && sliced_eligible) SPAM_TRY(sliced_analyse<T>(a, p->knobs, sl_cfg == 2, s, &q, want_stats ? &cs : nullptr));
    pc.stamp("sliced analysis (all of it)");
    if (q && sl_cfg == 1) {  // keep it only if it is faster on this matrix
      struct QGuard {
        SlicedPlan* q;
        ~QGuard() { if (q) sliced_plan_free(q); }
      } qg{q};
      q->ms_plain = ms_plain;
      SPAM_TRY(time_runs([&] { return sliced_run<T>(a, q, xt, (T*)ys.p, s); }, s, &q->ms_sliced));
      SPAM_HIP_TRY(hipStreamSynchronize(s));
      p->ms_plain = q->ms_plain, p->ms_sliced = q->ms_sliced;
      if (q->ms_sliced < 0.97f * q->ms_plain) {
        qg.q = nullptr;
      } else {
        q = nullptr;  // (freed by the guard)
      }
    }
    pc.stamp("timed comparison plain / sliced");
    p->sliced = q;
    if (q) {  // the plain path's hot-column table (4 bytes per entry) is not read again on this handle
      SPAM_HIP_TRY(hipStreamSynchronize(s));
      if (p->ja_hot) dev_free(p->ja_hot);
      if (p->hot_cols) dev_free(p->hot_cols);
      if (p->xh) dev_free(p->xh);
      p->ja_hot = nullptr, p->hot_cols = nullptr, p->xh = nullptr;  // (hot_k / hot_cover stay for spam_spmv_plan_info)
    }
  }
  p->prepared = true;
  guard.p = nullptr;
  __atomic_store_n(&a->spmv_plan, p, __ATOMIC_RELEASE);
  return SPAM_OK;
}

template <class T>
static int spmv_launch_t(const spam_csr* a, const void* d_x, void* d_y, hipStream_t s) {
  if (a->row_ptr32) return spmv_typed<T, uint32_t>(a, a->row_ptr32, (const T*)d_x, (T*)d_y, s);
  return spmv_typed<T, uint64_t>(a, a->row_ptr, (const T*)d_x, (T*)d_y, s);
}
int spmv_launch(const spam_csr* a, const void* d_x, void* d_y, hipStream_t s) {
  SPAM_TRY(wait_values(a, s));
  return SPAM_DISPATCH_DTYPE(a->dtype, spmv_launch_t<double>(a, d_x, d_y, s), spmv_launch_t<float>(a, d_x, d_y, s),
                             spmv_launch_t<i64w>(a, d_x, d_y, s));
}

template <class T>
static int spmv_prepare_t(spam_csr* a, const void* d_x, hipStream_t s) {
  if (a->row_ptr32) return prepare_typed<T, uint32_t>(a, a->row_ptr32, (const T*)d_x, s);
  return prepare_typed<T, uint64_t>(a, a->row_ptr, (const T*)d_x, s);
}
static int spmv_prepare(spam_csr* a, const void* d_x, hipStream_t s) {
  SPAM_TRY(wait_values(a, s));
  return SPAM_DISPATCH_DTYPE(a->dtype, spmv_prepare_t<double>(a, d_x, s), spmv_prepare_t<float>(a, d_x, s),
                             spmv_prepare_t<i64w>(a, d_x, s));
}

template <class T>
static int spmv_host(const spam_csr* a, int dtype, const T* x_host, T* y_host) {
  clear_error();
  if (!a || !x_host || !y_host) return set_error(SPAM_E_INVALID, "NULL argument");
  if (a->dtype != dtype) return set_error(SPAM_E_DTYPE, "handle dtype does not match entry point");
  hipStream_t s = stream();
  DevBuf dx, dy;
  SPAM_TRY(dx.alloc(a->cols * sizeof(T)));
  SPAM_TRY(dy.alloc(a->rows * sizeof(T)));
  SPAM_HIP_TRY(hipMemcpyAsync(dx.p, x_host, a->cols * sizeof(T), hipMemcpyHostToDevice, s));
  SPAM_TRY(spmv_launch(a, dx.p, dy.p, s));
  SPAM_HIP_TRY(hipMemcpyAsync(y_host, dy.p, a->rows * sizeof(T), hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  return SPAM_OK;
}

}  // namespace spam

using namespace spam;

extern "C" {

int spam_spmv_f64(const spam_csr* a, const double* x_host, double* y_host) {
  return spmv_host<double>(a, SPAM_F64, x_host, y_host);
}
int spam_spmv_f32(const spam_csr* a, const float* x_host, float* y_host) {
  return spmv_host<float>(a, SPAM_F32, x_host, y_host);
}
int spam_spmv_i64(const spam_csr* a, const int64_t* x_host, int64_t* y_host) {
  return spmv_host<i64w>(a, SPAM_I64, (const i64w*)x_host, (i64w*)y_host);
}
int spam_spmv_plan_info(const spam_csr* a, uint64_t* hot_columns, double* hot_cover, uint64_t* tiles) {
  clear_error();
  if (!a) return set_error(SPAM_E_INVALID, "matrix is NULL");
  const SpmvPlan* p = a->spmv_plan;
  if (hot_columns) *hot_columns = p ? p->hot_k : 0;
  if (hot_cover) *hot_cover = p ? p->hot_cover : 0.0;
  if (tiles) *tiles = p ? p->ntiles : 0;
  return SPAM_OK;
}

int spam_spmv_sliced_info(const spam_csr* a, int* active, uint64_t* long_rows, uint64_t* nnz_long, uint64_t* table_slots,
                          uint64_t* lds_slots, double* cover_table, double* cover_single, double* ms_plain,
                          double* ms_sliced) {
  clear_error();
  if (!a) return set_error(SPAM_E_INVALID, "matrix is NULL");
  const SlicedPlan* q = a->spmv_plan ? a->spmv_plan->sliced : nullptr;
  if (active) *active = q ? 1 : 0;
  if (long_rows) *long_rows = q ? q->nl : 0;
  if (nnz_long) *nnz_long = q ? q->nnz_long : 0;
  if (table_slots) *table_slots = q ? q->S : 0;
  if (lds_slots) *lds_slots = q ? q->L : 0;  // contiguous hottest slots (L1-resident; in LDS with SPAM_SPMV_LDS_KB)
  if (cover_table) *cover_table = q ? q->cover_table : 0.0;
  if (cover_single) *cover_single = q ? q->cover_single : 0.0;
  if (ms_plain) *ms_plain = a->spmv_plan ? a->spmv_plan->ms_plain : 0.0;
  if (ms_sliced) *ms_sliced = a->spmv_plan ? a->spmv_plan->ms_sliced : 0.0;
  return SPAM_OK;
}

int spam_spmv_rg_info(const spam_csr* a, int* active, uint64_t* row_groups, uint64_t* blocks, uint64_t* accumulators,
                      uint64_t* stream_bytes, uint64_t* plan_bytes) {
  clear_error();
  if (!a) return set_error(SPAM_E_INVALID, "matrix is NULL");
  const SlicedPlan* q = a->spmv_plan ? a->spmv_plan->sliced : nullptr;
  const RgPlan* g = q ? q->rg : nullptr;
  if (active) *active = g ? 1 : 0;
  if (row_groups) *row_groups = g ? g->ng : 0;
  if (blocks) *blocks = g ? g->nonempty_blocks : 0;
  if (accumulators) *accumulators = g ? g->naccs : 0;
  if (stream_bytes) *stream_bytes = g ? g->stream_bytes : 0;
  if (plan_bytes)
    *plan_bytes = g ? g->stream_bytes + (g->vsrc ? g->padded * 4 : 0) + ((uint64_t)g->ng * NPART + g->ngs + 1) * 48 + (uint64_t)g->ng * NPART * 8 +
                            (uint64_t)g->ntiles * (24 + 64 * 6) + g->npairs * 8 + (g->naccs - g->npairs) * 4
                      : 0;
  return SPAM_OK;
}

int spam_spmv_dev(const spam_csr* a, const void* d_x, void* d_y) {
  clear_error();
  if (!a || !d_x || !d_y) return set_error(SPAM_E_INVALID, "NULL argument");
  return spmv_launch(a, d_x, d_y, stream());
}

int spam_spmv_prepare(spam_csr* a, const void* d_x) {
  clear_error();
  if (!a) return set_error(SPAM_E_INVALID, "matrix is NULL");
  return spmv_prepare(a, d_x, stream());
}

int spam_csr_values_changed(spam_csr* a) {
  clear_error();
  if (!a) return set_error(SPAM_E_INVALID, "matrix is NULL");
  std::lock_guard<std::mutex> lk(a->plan_mu);
  SpmvPlan* p = a->spmv_plan;
  if (!p || !p->sliced) return SPAM_OK;  // the plain path reads the values where they are
  SPAM_TRY(wait_values(a, stream()));
  return SPAM_DISPATCH_DTYPE(a->dtype, sliced_refresh_values<double>(a, p->sliced, stream()),
                             sliced_refresh_values<float>(a, p->sliced, stream()),
                             sliced_refresh_values<i64w>(a, p->sliced, stream()));
}

}  // extern "C"
