// spam_spmv.hip — CSR SpMV y = A x for gfx950.
//
// Reference counterpart: none — the reference has no SpMV (its only product is mul_hash,
// spam_csr/src/lib.rs:292-297).  The result is DEFINED as A.mul_hash(X) with X = x as an n x 1 matrix
// (mul_hash.rs:145-163 with the single key 0): y_i = sum over row i in storage order,
// y_i = 0 for rows without entries.
//
// Design (HBM-bound: 12 B/nnz of ja/a stream + the x gather; no MFMA, nothing here is a dense contraction):
//   * the nnz range is cut into fixed tiles of TPB*NPT entries; workgroup t owns tile t, so every
//     workgroup streams the same number of bytes whatever the row-length distribution (R-MAT, 50k-entry
//     rows next to 8-entry rows) — this is the row "binning" of CSR-adaptive done per tile;
//   * phase 1: 16-byte coalesced loads of ja / a, gather x[ja], products into an LDS tile (padded one
//     slot per 32 so row-strided reads spread over the banks);
//   * phase 2, adaptive per row: rows of <= SHORT_ROW entries inside the tile are summed by one lane each,
//     left to right (bit-identical to the oracle's order); longer rows and the piece of a row that
//     started in an earlier tile are summed by a whole wavefront with __shfl_xor;
//   * a row spanning several tiles gets its pieces from `carry[tile]`, folded in tile order by a
//     second tiny kernel — deterministic, no float atomics.
//   The row -> tile map (first row starting in each tile) depends on row_ptr only and is built once per
//   matrix (the analysis step of CSR-adaptive), then reused by every SpMV on that handle.
#include <algorithm>
#include <cstdlib>

#include "spam_internal.hpp"

namespace spam {

struct SpmvPlan {
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
};

void spmv_plan_free(SpmvPlan* p) {
  if (!p) return;
  if (p->first_owned) dev_free(p->first_owned);
  if (p->carry) dev_free(p->carry);
  if (p->ja_hot) dev_free(p->ja_hot);
  if (p->hot_cols) dev_free(p->hot_cols);
  if (p->xh) dev_free(p->xh);
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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int pad_idx(int i) { return i + (i >> 5); }

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
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

template <class T, class OffT, int NPT, bool NT, int HOT>
__global__ __launch_bounds__(SPMV_TPB) void spmv_tile_kernel(const OffT* __restrict__ row_ptr,
                                                            const uint32_t* __restrict__ col,
                                                            const T* __restrict__ val, const T* __restrict__ x,
                                                            const T* __restrict__ xh, T* __restrict__ y,
                                                            const uint32_t* __restrict__ fo, T* __restrict__ carry,
                                                            uint64_t nnz) {
  constexpr int TILE = SPMV_TPB * NPT;
  constexpr int MAXLONG = TILE / (SHORT_ROW + 1) + 2;
  __shared__ T prod[TILE + TILE / 32 + 1];
  __shared__ uint32_t long_list[MAXLONG];
  __shared__ uint32_t n_long;

  const int tid = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * TILE;
  const uint32_t cnt = (uint32_t)((nnz - base) < (uint64_t)TILE ? (nnz - base) : (uint64_t)TILE);
  if (tid == 0) n_long = 0;

  // ---- phase 1: stream ja / a, gather x, products -> LDS ------------------------------------------
  if (cnt == TILE) {
    using VT = typename Vec4<T>::type;
    constexpr int VN = Vec4<T>::N;     // values per 16-byte load
    constexpr int NV = NPT / 4;        // 16-byte index loads per thread
    u32x4 c[NV];
    VT v[NV][4 / VN];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const uint64_t q = base + (uint64_t)(i * SPMV_TPB + tid) * 4;
      if (NT) {
        c[i] = __builtin_nontemporal_load((const u32x4*)(col + q));
#pragma unroll
        for (int j = 0; j < 4 / VN; ++j) v[i][j] = __builtin_nontemporal_load((const VT*)(val + q) + j);
      } else {
        c[i] = *(const u32x4*)(col + q);
#pragma unroll
        for (int j = 0; j < 4 / VN; ++j) v[i][j] = *((const VT*)(val + q) + j);
      }
    }
    T xv[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[i][j] = gather_x<T, HOT>(x, xh, c[i][j]);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = (i * SPMV_TPB + tid) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) prod[pad_idx(q + j)] = v[i][j / VN][j % VN] * xv[i][j];
    }
  } else {
    for (uint32_t q = tid; q < cnt; q += SPMV_TPB)
      prod[pad_idx(q)] = val[base + q] * gather_x<T, HOT>(x, xh, col[base + q]);
  }
  __syncthreads();

  // ---- phase 2: per-row sums -----------------------------------------------------------------------
  const uint32_t f0 = fo[blockIdx.x], f1 = fo[blockIdx.x + 1];
  const uint64_t end = base + cnt;
  const bool has_carry = f0 > 0 && (uint64_t)row_ptr[f0] > base;  // row f0-1 started before this tile
  if (has_carry && tid == 0) long_list[atomicAdd(&n_long, 1u)] = 0xFFFFFFFFu;
  for (uint32_t r = f0 + tid; r < f1; r += SPMV_TPB) {
    const uint64_t s = row_ptr[r];
    uint64_t e = row_ptr[r + 1];
    e = e < end ? e : end;
    const uint32_t len = (uint32_t)(e - s);
    if (len <= SHORT_ROW) {
      T acc = T(0);
      if (len) {
        int p = (int)(s - base);
        acc = prod[pad_idx(p)];
        for (uint32_t k = 1; k < len; ++k) acc += prod[pad_idx(p + (int)k)];
      }
      y[r] = acc;
    } else {
      long_list[atomicAdd(&n_long, 1u)] = r;
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t nl = n_long;
  for (uint32_t i = wave; i < nl; i += SPMV_TPB / 64) {
    const uint32_t r = long_list[i];
    uint64_t s, e;
    if (r == 0xFFFFFFFFu) {
      s = base;
      e = row_ptr[f0];
    } else {
      s = row_ptr[r];
      e = row_ptr[r + 1];
    }
    e = e < end ? e : end;
    T acc = T(0);
    for (int p = (int)(s - base) + lane, pe = (int)(e - base); p < pe; p += 64) acc += prod[pad_idx(p)];
    acc = wave_sum(acc);
    if (lane == 0) {
      if (r == 0xFFFFFFFFu)
        carry[blockIdx.x] = acc;
      else
        y[r] = acc;
    }
  }
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

// ---- hot-column analysis (once per matrix) ----------------------------------------------------------
__global__ void col_count_kernel(const uint32_t* __restrict__ col, uint64_t nnz, uint32_t* __restrict__ cnt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) atomicAdd(&cnt[col[i]], 1u);
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
template <class T>
static int spmv_hot_analyse(const spam_csr* a, SpmvPlan* p, int hot_kb, hipStream_t s) {
  p->hot_kb = hot_kb;
  p->hot_k = 0;
  const uint64_t kmax = (uint64_t)hot_kb * 1024 / sizeof(T);
  if (hot_kb <= 0 || a->cols >= HOT_BIT || a->cols <= 2 * kmax || a->nnz == 0) return SPAM_OK;
  const uint32_t cols = (uint32_t)a->cols;
  DevBuf cnt, out, flag, pos, tmp, slot;
  SPAM_TRY(cnt.alloc((uint64_t)cols * 4));
  SPAM_TRY(out.alloc(16));
  SPAM_HIP_TRY(hipMemsetAsync(cnt.p, 0, (uint64_t)cols * 4, s));
  col_count_kernel<<<8192, 256, 0, s>>>(a->col_idx, a->nnz, cnt.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  // bisect t in [1, 2^31): count(t) is non-increasing; want the smallest t with count(t) <= kmax
  uint32_t lo = 1, hi = 0x7FFFFFFFu;
  unsigned long long best[2] = {0, 0};
  const unsigned g = (unsigned)std::min<uint64_t>((cols + 255) / 256, 4096);
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    unsigned long long h[2];
    SPAM_HIP_TRY(hipMemsetAsync(out.p, 0, 16, s));
    count_ge_kernel<<<g, 256, 0, s>>>(cnt.as<uint32_t>(), cols, mid, out.as<unsigned long long>());
    SPAM_CHECK_LAUNCH();
    SPAM_HIP_TRY(hipMemcpyAsync(h, out.p, 16, hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
    if (h[0] <= kmax) {
      hi = mid;
      best[0] = h[0], best[1] = h[1];
    } else {
      lo = mid + 1;
    }
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

template <class T, class OffT>
static int spmv_typed(const spam_csr* a, const OffT* row_ptr, const T* x, T* y, hipStream_t s) {
  if (a->nnz == 0) {
    fill_zero_kernel<T><<<(unsigned)((a->rows + 255) / 256), 256, 0, s>>>(y, a->rows);
    SPAM_CHECK_LAUNCH();
    return SPAM_OK;
  }
  const int npt_cfg = env_int("SPAM_SPMV_NPT", 16);  // read per call so one process can A/B variants
  const int nt_cfg = env_int("SPAM_SPMV_NT", 1);  // non-temporal ja/a stream: keeps L2 for x (+4..6 % on random gathers)
  const int npt = (npt_cfg == 8) ? 8 : 16;
  const uint32_t tile = SPMV_TPB * npt;
  spam_csr* am = const_cast<spam_csr*>(a);
  if (!am->spmv_plan || am->spmv_plan->npt != npt) {
    if (am->spmv_plan) spmv_plan_free(am->spmv_plan);
    am->spmv_plan = nullptr;
    SpmvPlan* p = new (std::nothrow) SpmvPlan();
    if (!p) return set_error(SPAM_E_OOM, "host allocation failed");
    p->npt = npt;
    p->ntiles = (a->nnz + tile - 1) / tile;
    DevBuf fo, ca;
    if (int e = fo.alloc((p->ntiles + 1) * 4)) {
      delete p;
      return e;
    }
    if (int e = ca.alloc(p->ntiles * sizeof(T))) {
      delete p;
      return e;
    }
    p->first_owned = (uint32_t*)fo.release();
    p->carry = ca.release();
    spmv_first_owned_kernel<OffT><<<(unsigned)((p->ntiles + 1 + 255) / 256), 256, 0, s>>>(
        row_ptr, (uint32_t)a->rows, tile, p->ntiles, p->first_owned);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
      spmv_plan_free(p);
      return set_error(SPAM_E_HIP, "spmv_first_owned_kernel: %s", hipGetErrorString(le));
    }
    am->spmv_plan = p;
  }
  const int hot_cfg = env_int("SPAM_SPMV_HOT", 2);        // 0 off, 1 per-lane base select, 2 cold gathers non-temporal
  const int hot_kb = hot_cfg ? env_int("SPAM_SPMV_HOT_KB", 3072) : 0;  // of the 4 MB L2 per XCD
  if (am->spmv_plan->hot_kb != hot_kb) {
    SpmvPlan* pm = am->spmv_plan;
    if (pm->ja_hot) dev_free(pm->ja_hot);
    if (pm->hot_cols) dev_free(pm->hot_cols);
    if (pm->xh) dev_free(pm->xh);
    pm->ja_hot = pm->hot_cols = nullptr;
    pm->xh = nullptr;
    SPAM_TRY(spmv_hot_analyse<T>(a, pm, hot_kb, s));
  }
  const SpmvPlan* p = a->spmv_plan;
  const unsigned grid = (unsigned)p->ntiles;
  const int hot = p->hot_k ? hot_cfg : 0;
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

int spmv_launch(const spam_csr* a, const void* d_x, void* d_y, hipStream_t s) {
  if (a->dtype == SPAM_F64) {
    if (a->row_ptr32) return spmv_typed<double, uint32_t>(a, a->row_ptr32, (const double*)d_x, (double*)d_y, s);
    return spmv_typed<double, uint64_t>(a, a->row_ptr, (const double*)d_x, (double*)d_y, s);
  }
  if (a->row_ptr32) return spmv_typed<float, uint32_t>(a, a->row_ptr32, (const float*)d_x, (float*)d_y, s);
  return spmv_typed<float, uint64_t>(a, a->row_ptr, (const float*)d_x, (float*)d_y, s);
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
int spam_spmv_plan_info(const spam_csr* a, uint64_t* hot_columns, double* hot_cover, uint64_t* tiles) {
  clear_error();
  if (!a) return set_error(SPAM_E_INVALID, "matrix is NULL");
  const SpmvPlan* p = a->spmv_plan;
  if (hot_columns) *hot_columns = p ? p->hot_k : 0;
  if (hot_cover) *hot_cover = p ? p->hot_cover : 0.0;
  if (tiles) *tiles = p ? p->ntiles : 0;
  return SPAM_OK;
}

int spam_spmv_dev(const spam_csr* a, const void* d_x, void* d_y) {
  clear_error();
  if (!a || !d_x || !d_y) return set_error(SPAM_E_INVALID, "NULL argument");
  return spmv_launch(a, d_x, d_y, stream());
}

}  // extern "C"
