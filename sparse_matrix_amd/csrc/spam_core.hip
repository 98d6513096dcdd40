// spam_core.hip — runtime, error reporting and CsrMatrix <-> HBM movement for libspam_hip.so.
//
// Reference counterpart: the CsrMatrix container itself (spam_csr/src/lib.rs:25-32) and its invariants
// (lib.rs:47-81).  Nothing here is arithmetic; the kernels only narrow/widen indices (usize <-> u32, the
// truncation the reference performs per product at mul_hash.rs:92,157 happens ONCE here) and validate.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <time.h>

#include "spam_internal.hpp"

namespace spam {

static thread_local std::string g_err;
static hipStream_t g_stream = nullptr;

int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
void clear_error() { g_err.clear(); }
double PhaseClock::now() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

hipStream_t stream() { return g_stream; }

// ---- caching device allocator ---------------------------------------------------------------------
static std::mutex g_pool_mu;
static std::multimap<size_t, void*> g_pool_free;        // size -> block
static std::unordered_map<void*, size_t> g_pool_sizes;  // every live block handed out or cached

int dev_alloc(void** out, size_t bytes) {
  const size_t n = (bytes + 255) & ~(size_t)255;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_free.lower_bound(n);
    if (it != g_pool_free.end() && it->first <= n + n / 4 + (1u << 20)) {  // at most 25 % (+1 MiB) slack
      *out = it->second;
      g_pool_free.erase(it);
      return SPAM_OK;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    dev_trim();  // give cached blocks back and retry once
    e = hipMalloc(&p, n);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return set_error(SPAM_E_OOM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool_sizes[p] = n;
  *out = p;
  return SPAM_OK;
}

void dev_free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  auto it = g_pool_sizes.find(p);
  if (it == g_pool_sizes.end()) {
    (void)hipFree(p);
    return;
  }
  g_pool_free.emplace(it->second, p);
}

void dev_trim() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto& kv : g_pool_free) {
    g_pool_sizes.erase(kv.second);
    (void)hipFree(kv.second);
  }
  g_pool_free.clear();
}

// ---- index narrowing / widening ------------------------------------------------------------------
__global__ void narrow_u64_u32_kernel(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                      uint64_t limit, int* __restrict__ bad) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  int b = 0;
  for (; i < n; i += stride) {
    uint64_t v = in[i];
    b |= (v >= limit);
    out[i] = (uint32_t)v;
  }
  if (b) *bad = 1;
}
__global__ void widen_u32_u64_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

int narrow_row_ptr(const uint64_t* d_in, uint64_t n, uint32_t* d_out, hipStream_t s) {
  if (n == 0) return SPAM_OK;
  DevBuf bad;
  SPAM_TRY(bad.alloc(sizeof(int)));
  SPAM_HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(int), s));
  unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
  narrow_u64_u32_kernel<<<grid, 256, 0, s>>>(d_in, d_out, n, (uint64_t)1 << 32, bad.as<int>());
  SPAM_CHECK_LAUNCH();
  int h = 0;
  SPAM_HIP_TRY(hipMemcpyAsync(&h, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (h) return set_error(SPAM_E_INVALID, "row_ptr entry >= 2^32 while nnz < 2^32");
  return SPAM_OK;
}

// ---- invariants on the device (spam_csr/src/lib.rs:47-81) ----------------------------------------
// One wave per row.  bit 0: offsets not sorted (3); bit 1: index >= cols (5); bit 2: row not strictly
// increasing (6, IS_SORTED) / has a duplicate (6, unsorted).  Unsorted rows longer than 4096 go to invariant6_long_kernel.
__global__ void invariants_kernel(const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, uint64_t rows,
                                  uint64_t cols, uint64_t nnz, int is_sorted, int* __restrict__ bad,
                                  uint32_t* __restrict__ long_rows, uint32_t* __restrict__ n_long) {
  const int lane = threadIdx.x & 63;
  uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  int b = 0;
  for (uint64_t r = wave; r < rows; r += nwaves) {
    uint64_t s = row_ptr[r], e = row_ptr[r + 1];
    if (e < s || e > nnz) {
      b |= 1;
      continue;
    }
    for (uint64_t p = s + lane; p < e; p += 64) {
      uint32_t c = col[p];
      if (c >= cols) b |= 2;
      if (is_sorted) {
        if (p > s && col[p - 1] >= c) b |= 4;
      }
    }
    if (!is_sorted) {
      uint64_t len = e - s;
      if (len > 4096) {
        if (lane == 0) long_rows[atomicAdd(n_long, 1u)] = (uint32_t)r;  // checked by invariant6_long_kernel
      } else {
        for (uint64_t i = 0; i < len; ++i) {
          uint32_t ci = col[s + i];
          for (uint64_t j = i + 1 + lane; j < len; j += 64)
            if (col[s + j] == ci) b |= 4;
        }
      }
    }
  }
  if (b) atomicOr(bad, b);
}

// Invariant 6 for long unsorted rows: one workgroup per row, a 2^20-column bitmap window in LDS;
// atomicOr returning an already-set bit means a duplicate column.
__global__ __launch_bounds__(1024) void invariant6_long_kernel(const uint64_t* __restrict__ row_ptr,
                                                              const uint32_t* __restrict__ col,
                                                              const uint32_t* __restrict__ long_rows, uint64_t cols,
                                                              int* __restrict__ bad) {
  __shared__ uint32_t bm[32768];  // 2^20 bits
  const uint32_t r = long_rows[blockIdx.x];
  const uint64_t s = row_ptr[r], e = row_ptr[r + 1];
  int b = 0;
  for (uint64_t w0 = 0; w0 < cols; w0 += (1u << 20)) {
    for (int i = threadIdx.x; i < 32768; i += 1024) bm[i] = 0;
    __syncthreads();
    for (uint64_t p = s + threadIdx.x; p < e; p += 1024) {
      const uint64_t c = (uint64_t)col[p] - w0;
      if (c < (1u << 20)) {
        const uint32_t bit = 1u << (c & 31);
        if (atomicOr(&bm[c >> 5], bit) & bit) b = 4;
      }
    }
    __syncthreads();
  }
  if (b) atomicOr(bad, b);
}

// float4 grid-stride copy: the measured ceiling for streaming kernels on this device (spam_hip_copy_gbps)
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

static int check_aligned(const void* p, const char* what) {
  if (((uintptr_t)p & 15) != 0) return set_error(SPAM_E_INVALID, "%s is not 16-byte aligned", what);
  return SPAM_OK;
}

static int upload_impl(int dtype, uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* offsets,
                       const uint64_t* indices, const void* vals, int is_sorted, spam_csr** out) {
  clear_error();
  if (!out) return set_error(SPAM_E_INVALID, "out is NULL");
  *out = nullptr;
  if (rows == 0 || cols == 0) return set_error(SPAM_E_INVALID, "rows and cols are NonZeroUsize (lib.rs:27-28)");
  if (!offsets || (nnz && (!indices || !vals))) return set_error(SPAM_E_INVALID, "NULL array");
  if (cols >= 0xFFFFFFFFull)
    return set_error(SPAM_E_INDEX_RANGE, "cols = %llu: column keys must be < u32::MAX (mul_hash.rs:12)",
                     (unsigned long long)cols);
  if (rows >= 0xFFFFFFFFull) return set_error(SPAM_E_UNSUPPORTED, "rows >= 2^32-1 not supported");
  // invariants 7, 3, 4 on the host copy (the reference would panic later on a malformed matrix)
  if (offsets[0] != 0) return set_error(SPAM_E_INVALID, "offsets[0] != 0 (invariant 7)");
  for (uint64_t r = 0; r < rows; ++r)
    if (offsets[r + 1] < offsets[r]) return set_error(SPAM_E_INVALID, "offsets not sorted at row %llu (invariant 3)", (unsigned long long)r);
  if (offsets[rows] != nnz) return set_error(SPAM_E_INVALID, "offsets[rows] != nnz (invariant 4)");

  hipStream_t s = stream();
  spam_csr* m = new (std::nothrow) spam_csr();
  if (!m) return set_error(SPAM_E_OOM, "host allocation failed");
  m->rows = rows, m->cols = cols, m->nnz = nnz, m->dtype = dtype, m->is_sorted = is_sorted;
  auto fail = [&](int code) {
    spam_csr_free(m);
    return code;
  };
  const size_t vs = dtype_size(dtype);
  DevBuf rp, ci, va;
  if (int e = rp.alloc((rows + 1) * 8)) return fail(e);
  if (int e = ci.alloc(nnz * 4)) return fail(e);
  if (int e = va.alloc(nnz * vs)) return fail(e);
  m->row_ptr = (uint64_t*)rp.release();
  m->col_idx = (uint32_t*)ci.release();
  m->vals = va.release();
  m->owns_row_ptr = m->owns_col_val = true;
#define UP_TRY(expr)                                                                             \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) return fail(set_error(SPAM_E_HIP, "%s: %s", #expr, hipGetErrorString(_e))); \
  } while (0)
  UP_TRY(hipMemcpyAsync(m->row_ptr, offsets, (rows + 1) * 8, hipMemcpyHostToDevice, s));
  if (nnz) UP_TRY(hipMemcpyAsync(m->vals, vals, nnz * vs, hipMemcpyHostToDevice, s));
  // indices: stage as u64 in chunks, narrow on the device, flag any index >= cols (invariant 5)
  const uint64_t CH = 1ull << 25;
  DevBuf stage, bad;
  if (int e = stage.alloc(std::min(nnz, CH) * 8)) return fail(e);
  if (int e = bad.alloc(sizeof(int))) return fail(e);
  UP_TRY(hipMemsetAsync(bad.p, 0, sizeof(int), s));
  for (uint64_t o = 0; o < nnz; o += CH) {
    uint64_t n = std::min(CH, nnz - o);
    UP_TRY(hipMemcpyAsync(stage.p, indices + o, n * 8, hipMemcpyHostToDevice, s));
    unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
    narrow_u64_u32_kernel<<<grid, 256, 0, s>>>(stage.as<uint64_t>(), m->col_idx + o, n, cols, bad.as<int>());
    UP_TRY(hipGetLastError());
    UP_TRY(hipStreamSynchronize(s));  // stage is reused
  }
  int hbad = 0;
  UP_TRY(hipMemcpyAsync(&hbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
  UP_TRY(hipStreamSynchronize(s));
  if (hbad) return fail(set_error(SPAM_E_INVALID, "column index >= cols (invariant 5)"));
  if (nnz < (1ull << 32)) {
    DevBuf r32;
    if (int e = r32.alloc((rows + 1) * 4)) return fail(e);
    m->row_ptr32 = (uint32_t*)r32.release();
    m->owns_row_ptr32 = true;
    if (int e = narrow_row_ptr(m->row_ptr, rows + 1, m->row_ptr32, s)) return fail(e);
  }
#undef UP_TRY
  *out = m;
  return SPAM_OK;
}

template <class T>
static int download_impl(const spam_csr* m, int dtype, uint64_t* offsets, uint64_t* indices, T* vals) {
  clear_error();
  if (!m) return set_error(SPAM_E_INVALID, "matrix is NULL");
  if (m->dtype != dtype) return set_error(SPAM_E_DTYPE, "handle dtype does not match entry point");
  hipStream_t s = stream();
  SPAM_TRY(wait_values(m, s));
  if (offsets) SPAM_HIP_TRY(hipMemcpyAsync(offsets, m->row_ptr, (m->rows + 1) * 8, hipMemcpyDeviceToHost, s));
  if (vals && m->nnz) SPAM_HIP_TRY(hipMemcpyAsync(vals, m->vals, m->nnz * sizeof(T), hipMemcpyDeviceToHost, s));
  if (indices && m->nnz) {
    const uint64_t CH = 1ull << 25;
    DevBuf stage;
    SPAM_TRY(stage.alloc(std::min(m->nnz, CH) * 8));
    for (uint64_t o = 0; o < m->nnz; o += CH) {
      uint64_t n = std::min(CH, m->nnz - o);
      unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
      widen_u32_u64_kernel<<<grid, 256, 0, s>>>(m->col_idx + o, stage.as<uint64_t>(), n);
      SPAM_CHECK_LAUNCH();
      SPAM_HIP_TRY(hipMemcpyAsync(indices + o, stage.p, n * 8, hipMemcpyDeviceToHost, s));
      SPAM_HIP_TRY(hipStreamSynchronize(s));
    }
  }
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  return SPAM_OK;
}

}  // namespace spam

using namespace spam;

extern "C" {

int spam_version(void) { return 1002; }  // 1001: + spmv_prepare, values_changed, rank_step, i64 entry points, copy_gbps; 1002: + row_flops_dev, rows_to_parts_from_flops_dev, libspam_hip_rccl.so

const char* spam_last_error(void) { return g_err.c_str(); }

int spam_hip_init(int device) {
  clear_error();
  int n = 0;
  SPAM_HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return set_error(SPAM_E_INVALID, "device %d out of range (%d visible)", device, n);
  SPAM_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t p;
  SPAM_HIP_TRY(hipGetDeviceProperties(&p, device));
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
    return set_error(SPAM_E_UNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only", device,
                     p.gcnArchName);
  return SPAM_OK;
}

int spam_hip_set_stream(void* hip_stream) {
  clear_error();
  g_stream = (hipStream_t)hip_stream;
  return SPAM_OK;
}

int spam_hip_trim(void) {
  clear_error();
  SPAM_HIP_TRY(hipStreamSynchronize(stream()));
  dev_trim();
  return SPAM_OK;
}

int spam_hip_copy_gbps(uint64_t bytes, int reps, double* gbps_out) {
  clear_error();
  if (!gbps_out || bytes < 16 || reps < 1) return set_error(SPAM_E_INVALID, "bad argument");
  hipStream_t s = stream();
  const uint64_t n = bytes / 16;
  DevBuf a, b;
  SPAM_TRY(a.alloc(n * 16));
  SPAM_TRY(b.alloc(n * 16));
  SPAM_HIP_TRY(hipMemsetAsync(a.p, 1, n * 16, s));
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 65536);  // (tools/copy_bench.hip: the fastest of the variants tried)
  hipEvent_t e0, e1;
  SPAM_HIP_TRY(hipEventCreate(&e0));
  SPAM_HIP_TRY(hipEventCreate(&e1));
  stream_copy_kernel<<<grid, 256, 0, s>>>(a.as<float4>(), b.as<float4>(), n);  // warm-up
  (void)hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) stream_copy_kernel<<<grid, 256, 0, s>>>(a.as<float4>(), b.as<float4>(), n);
  (void)hipEventRecord(e1, s);
  hipError_t le = hipGetLastError();
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (le != hipSuccess) return set_error(SPAM_E_HIP, "stream_copy_kernel: %s", hipGetErrorString(le));
  *gbps_out = 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9;  // read + write
  return SPAM_OK;
}

int spam_hip_synchronize(void) {
  clear_error();
  SPAM_HIP_TRY(hipStreamSynchronize(stream()));
  return SPAM_OK;
}

int spam_csr_upload_f64(uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* offsets, const uint64_t* indices,
                        const double* vals, int is_sorted, spam_csr** out) {
  return upload_impl(SPAM_F64, rows, cols, nnz, offsets, indices, vals, is_sorted, out);
}
int spam_csr_upload_f32(uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* offsets, const uint64_t* indices,
                        const float* vals, int is_sorted, spam_csr** out) {
  return upload_impl(SPAM_F32, rows, cols, nnz, offsets, indices, vals, is_sorted, out);
}
int spam_csr_upload_i64(uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* offsets, const uint64_t* indices,
                        const int64_t* vals, int is_sorted, spam_csr** out) {
  return upload_impl(SPAM_I64, rows, cols, nnz, offsets, indices, vals, is_sorted, out);
}

int spam_csr_from_device(int dtype, uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* d_row_ptr,
                         const uint32_t* d_col_idx, const void* d_vals, int is_sorted, spam_csr** out) {
  clear_error();
  if (!out) return set_error(SPAM_E_INVALID, "out is NULL");
  *out = nullptr;
  if (dtype != SPAM_F32 && dtype != SPAM_F64 && dtype != SPAM_I64) return set_error(SPAM_E_INVALID, "bad dtype %d", dtype);
  if (rows == 0 || cols == 0) return set_error(SPAM_E_INVALID, "rows and cols are NonZeroUsize (lib.rs:27-28)");
  if (cols >= 0xFFFFFFFFull) return set_error(SPAM_E_INDEX_RANGE, "cols must be < u32::MAX (mul_hash.rs:12)");
  if (rows >= 0xFFFFFFFFull) return set_error(SPAM_E_UNSUPPORTED, "rows >= 2^32-1 not supported");
  if (!d_row_ptr || (nnz && (!d_col_idx || !d_vals))) return set_error(SPAM_E_INVALID, "NULL device array");
  SPAM_TRY(check_aligned(d_row_ptr, "d_row_ptr"));
  SPAM_TRY(check_aligned(d_col_idx, "d_col_idx"));
  SPAM_TRY(check_aligned(d_vals, "d_vals"));
  spam_csr* m = new (std::nothrow) spam_csr();
  if (!m) return set_error(SPAM_E_OOM, "host allocation failed");
  m->rows = rows, m->cols = cols, m->nnz = nnz, m->dtype = dtype, m->is_sorted = is_sorted;
  m->row_ptr = const_cast<uint64_t*>(d_row_ptr);
  m->col_idx = const_cast<uint32_t*>(d_col_idx);
  m->vals = const_cast<void*>(d_vals);
  if (nnz < (1ull << 32)) {
    DevBuf r32;
    if (int e = r32.alloc((rows + 1) * 4)) {
      delete m;
      return e;
    }
    m->row_ptr32 = (uint32_t*)r32.release();
    m->owns_row_ptr32 = true;
    if (int e = narrow_row_ptr(m->row_ptr, rows + 1, m->row_ptr32, stream())) {
      spam_csr_free(m);
      return e;
    }
  }
  *out = m;
  return SPAM_OK;
}

__global__ void rebase_row_ptr_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                      uint32_t* __restrict__ out32, uint64_t n, uint64_t base) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t v = in[i] - base;
    out[i] = v;
    if (out32) out32[i] = (uint32_t)v;
  }
}

// row_ptr from the row indices of (row, col)-sorted triples: entry i opens rows (row[i-1], row[i]]
__global__ void coo_row_ptr_kernel(const uint32_t* __restrict__ row, uint64_t nnz, uint64_t rows,
                                   uint64_t* __restrict__ row_ptr, int* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nnz) return;
  const uint64_t hi = i < nnz ? row[i] : rows;              // sentinel entry closes the trailing rows
  const uint64_t lo = i == 0 ? 0 : (uint64_t)row[i - 1] + 1;  // first row not yet opened
  if (i < nnz && (hi >= rows || (i > 0 && row[i] < row[i - 1]))) {
    *bad = 1;
    return;
  }
  for (uint64_t r = (i == 0 ? 0 : lo); r <= hi && r <= rows; ++r) row_ptr[r] = i;
}

int spam_csr_from_sorted_coo(int dtype, uint64_t rows, uint64_t cols, uint64_t nnz, const uint32_t* d_row_idx,
                             const uint32_t* d_col_idx, const void* d_vals, int is_sorted, spam_csr** out) {
  clear_error();
  if (!out) return set_error(SPAM_E_INVALID, "out is NULL");
  *out = nullptr;
  if (nnz && !d_row_idx) return set_error(SPAM_E_INVALID, "NULL device array");
  hipStream_t s = stream();
  DevBuf rp, bad;
  SPAM_TRY(rp.alloc((rows + 1) * 8));
  SPAM_TRY(bad.alloc(sizeof(int)));
  SPAM_HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(int), s));
  coo_row_ptr_kernel<<<(unsigned)((nnz + 1 + 255) / 256), 256, 0, s>>>(d_row_idx, nnz, rows, rp.as<uint64_t>(), bad.as<int>());
  SPAM_CHECK_LAUNCH();
  int h = 0;
  SPAM_HIP_TRY(hipMemcpyAsync(&h, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (h) return set_error(SPAM_E_INVALID, "row indices are not sorted or >= rows");
  SPAM_TRY(spam_csr_from_device(dtype, rows, cols, nnz, rp.as<uint64_t>(), d_col_idx, d_vals, is_sorted, out));
  (*out)->row_ptr = (uint64_t*)rp.release();  // the handle owns the row_ptr it was built with
  (*out)->owns_row_ptr = true;
  return SPAM_OK;
}

int spam_csr_row_block(const spam_csr* a, uint64_t row_lo, uint64_t row_hi, spam_csr** out) {
  clear_error();
  if (!a || !out) return set_error(SPAM_E_INVALID, "NULL argument");
  *out = nullptr;
  if (row_lo >= row_hi || row_hi > a->rows)
    return set_error(SPAM_E_INVALID, "row block [%llu, %llu) is empty or outside 0..%llu", (unsigned long long)row_lo,
                     (unsigned long long)row_hi, (unsigned long long)a->rows);
  hipStream_t s = stream();
  uint64_t ends[2];
  SPAM_HIP_TRY(hipMemcpyAsync(&ends[0], a->row_ptr + row_lo, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(&ends[1], a->row_ptr + row_hi, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  // keep the 16-byte alignment the vector loads rely on: start the view at a multiple of 4 entries
  uint64_t base = ends[0] & ~3ull;
  spam_csr* m = new (std::nothrow) spam_csr();
  if (!m) return set_error(SPAM_E_OOM, "host allocation failed");
  m->rows = row_hi - row_lo, m->cols = a->cols, m->dtype = a->dtype, m->is_sorted = a->is_sorted;
  m->nnz = ends[1] - base;  // the view's arrays start at `base`; entries [base, ends[0]) belong to no row
  m->col_idx = a->col_idx + base;
  m->vals = (char*)a->vals + base * dtype_size(a->dtype);
  m->vals_ready = a->vals_ready;  // a view of a matrix whose values are still arriving shares its event (not owned)
  DevBuf rp, rp32;
  if (int e = rp.alloc((m->rows + 1) * 8)) {
    delete m;
    return e;
  }
  bool want32 = m->nnz < (1ull << 32);
  if (want32)
    if (int e = rp32.alloc((m->rows + 1) * 4)) {
      delete m;
      return e;
    }
  unsigned grid = (unsigned)std::min<uint64_t>((m->rows + 256) / 256, 4096);
  rebase_row_ptr_kernel<<<grid, 256, 0, s>>>(a->row_ptr + row_lo, rp.as<uint64_t>(), want32 ? rp32.as<uint32_t>() : nullptr,
                                             m->rows + 1, base);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    delete m;
    return set_error(SPAM_E_HIP, "rebase_row_ptr_kernel: %s", hipGetErrorString(le));
  }
  m->row_ptr = (uint64_t*)rp.release();
  m->owns_row_ptr = true;
  if (want32) {
    m->row_ptr32 = (uint32_t*)rp32.release();
    m->owns_row_ptr32 = true;
  }
  *out = m;
  return SPAM_OK;
}

int spam_csr_dims(const spam_csr* m, uint64_t* rows, uint64_t* cols, uint64_t* nnz, int* dtype, int* is_sorted) {
  clear_error();
  if (!m) return set_error(SPAM_E_INVALID, "matrix is NULL");
  if (rows) *rows = m->rows;
  if (cols) *cols = m->cols;
  if (nnz) *nnz = m->nnz;
  if (dtype) *dtype = m->dtype;
  if (is_sorted) *is_sorted = m->is_sorted;
  return SPAM_OK;
}

int spam_csr_device_ptrs(const spam_csr* m, const uint64_t** d_row_ptr, const uint32_t** d_col_idx,
                         const void** d_vals) {
  clear_error();
  if (!m) return set_error(SPAM_E_INVALID, "matrix is NULL");
  if (m->vals_ready && d_vals) SPAM_HIP_TRY(hipEventSynchronize(m->vals_ready));  // raw pointers leave the library: values must be there
  if (d_row_ptr) *d_row_ptr = m->row_ptr;
  if (d_col_idx) *d_col_idx = m->col_idx;
  if (d_vals) *d_vals = m->vals;
  return SPAM_OK;
}

int spam_csr_download_f64(const spam_csr* m, uint64_t* offsets, uint64_t* indices, double* vals) {
  return download_impl<double>(m, SPAM_F64, offsets, indices, vals);
}
int spam_csr_download_f32(const spam_csr* m, uint64_t* offsets, uint64_t* indices, float* vals) {
  return download_impl<float>(m, SPAM_F32, offsets, indices, vals);
}
int spam_csr_download_i64(const spam_csr* m, uint64_t* offsets, uint64_t* indices, int64_t* vals) {
  return download_impl<int64_t>(m, SPAM_I64, offsets, indices, vals);
}

int spam_csr_invariants(const spam_csr* m, int* ok) {
  clear_error();
  if (!m || !ok) return set_error(SPAM_E_INVALID, "NULL argument");
  *ok = 0;
  hipStream_t s = stream();
  uint64_t ends[2];
  SPAM_HIP_TRY(hipMemcpyAsync(&ends[0], m->row_ptr, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(&ends[1], m->row_ptr + m->rows, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  // a row-block view starts its arrays up to 3 entries early (alignment), so offsets[0] may be 0..3 there
  bool is_view = !m->owns_col_val && m->owns_row_ptr;
  if ((is_view ? ends[0] > 3 : ends[0] != 0) || ends[1] != m->nnz) return SPAM_OK;  // invariants 7 / 4
  DevBuf bad, long_rows;
  SPAM_TRY(bad.alloc(2 * sizeof(int)));
  SPAM_TRY(long_rows.alloc(m->is_sorted ? 4 : m->rows * 4));
  SPAM_HIP_TRY(hipMemsetAsync(bad.p, 0, 2 * sizeof(int), s));
  unsigned grid = (unsigned)std::min<uint64_t>((m->rows + 3) / 4, 8192);
  invariants_kernel<<<grid, 256, 0, s>>>(m->row_ptr, m->col_idx, m->rows, m->cols, m->nnz, m->is_sorted, bad.as<int>(),
                                         long_rows.as<uint32_t>(), bad.as<uint32_t>() + 1);
  SPAM_CHECK_LAUNCH();
  int h[2] = {0, 0};
  SPAM_HIP_TRY(hipMemcpyAsync(h, bad.p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (h[0] == 0 && h[1] > 0) {
    invariant6_long_kernel<<<(unsigned)h[1], 1024, 0, s>>>(m->row_ptr, m->col_idx, long_rows.as<uint32_t>(), m->cols,
                                                          bad.as<int>());
    SPAM_CHECK_LAUNCH();
    SPAM_HIP_TRY(hipMemcpyAsync(h, bad.p, sizeof(int), hipMemcpyDeviceToHost, s));
    SPAM_HIP_TRY(hipStreamSynchronize(s));
  }
  *ok = (h[0] == 0);
  return SPAM_OK;
}

void spam_csr_free(spam_csr* m) {
  if (!m) return;
  if (m->vals_ready && m->owns_vals_event) {  // the exchange stream may still be writing vals: do not recycle them under it
    (void)hipEventSynchronize(m->vals_ready);
    (void)hipEventDestroy(m->vals_ready);
  }
  if (m->spmv_plan) spmv_plan_free(m->spmv_plan);
  if (m->owns_row_ptr && m->row_ptr) dev_free(m->row_ptr);
  if (m->owns_row_ptr32 && m->row_ptr32) dev_free(m->row_ptr32);
  if (m->owns_col_val) {
    if (m->col_idx) dev_free(m->col_idx);
    if (m->vals) dev_free(m->vals);
  }
  delete m;
}

}  // extern "C"
