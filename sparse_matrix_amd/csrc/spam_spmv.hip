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
#include <cstdlib>

#include "spam_internal.hpp"

namespace spam {

struct SpmvPlan {
  uint32_t* first_owned = nullptr;  // [ntiles + 1]: first row whose start lies in tile t (rows for t = ntiles)
  void* carry = nullptr;            // T[ntiles]
  uint64_t ntiles = 0;
  int npt = 0;                      // nnz per thread the map was built for
};

void spmv_plan_free(SpmvPlan* p) {
  if (!p) return;
  if (p->first_owned) dev_free(p->first_owned);
  if (p->carry) dev_free(p->carry);
  delete p;
}

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

template <class T, class OffT, int NPT, bool NT>
__global__ __launch_bounds__(SPMV_TPB) void spmv_tile_kernel(const OffT* __restrict__ row_ptr,
                                                            const uint32_t* __restrict__ col,
                                                            const T* __restrict__ val, const T* __restrict__ x,
                                                            T* __restrict__ y, const uint32_t* __restrict__ fo,
                                                            T* __restrict__ carry, uint64_t nnz) {
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
      for (int j = 0; j < 4; ++j) xv[i][j] = x[c[i][j]];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = (i * SPMV_TPB + tid) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) prod[pad_idx(q + j)] = v[i][j / VN][j % VN] * xv[i][j];
    }
  } else {
    for (uint32_t q = tid; q < cnt; q += SPMV_TPB) prod[pad_idx(q)] = val[base + q] * x[col[base + q]];
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
  const int nt_cfg = env_int("SPAM_SPMV_NT", 0);
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
  const SpmvPlan* p = a->spmv_plan;
  const unsigned grid = (unsigned)p->ntiles;
  const uint32_t* col = a->col_idx;
  const T* val = (const T*)a->vals;
  T* carry = (T*)p->carry;
#define SPAM_LAUNCH_TILE(NPT_, NT_)                                                                             \
  spmv_tile_kernel<T, OffT, NPT_, NT_><<<grid, SPMV_TPB, 0, s>>>(row_ptr, col, val, x, y, p->first_owned, carry, \
                                                                  a->nnz)
  if (npt == 8) {
    if (nt_cfg) SPAM_LAUNCH_TILE(8, true); else SPAM_LAUNCH_TILE(8, false);
  } else {
    if (nt_cfg) SPAM_LAUNCH_TILE(16, true); else SPAM_LAUNCH_TILE(16, false);
  }
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
int spam_spmv_dev(const spam_csr* a, const void* d_x, void* d_y) {
  clear_error();
  if (!a || !d_x || !d_y) return set_error(SPAM_E_INVALID, "NULL argument");
  return spmv_launch(a, d_x, d_y, stream());
}

}  // extern "C"
