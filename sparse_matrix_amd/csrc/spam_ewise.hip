// spam_ewise.hip — element-wise C = A + B / A - B on CSR (off the SpMV/SpGEMM hot path, the
// next most used operation of the reference).
//
// Reference counterpart: CsrMatrix::apply_elementwise + impl Add / Sub (spam_csr/src/lib.rs:83-149, 276-290).
// For IS_SORTED inputs the reference merge-joins the two rows by column (lib.rs:101-118): the result row is the
// sorted union, an entry present in both gets f(a, b), one present only in A f(a, 0), only in B f(0, b); results
// that happen to be zero are KEPT (nothing prunes them).  Unsorted inputs go through a std HashMap whose iteration
// order is unspecified (lib.rs:119-134): here they are sorted first (private copies) and the result is the sorted
// union — the same SET per row, which is all the reference guarantees.
//
// Two passes like SpGEMM: count (|A_r| + |B_r| - |A_r ∩ B_r| by binary search), scan, fill.  One wavefront per row;
// an entry's output position = #A entries before + #B entries before - #common entries before, all from binary
// searches and one wave prefix sum, so the fill needs no atomics and the output is deterministic.
// HBM-bound: reads (12 B/nnz of A and B) twice, writes 12 B/nnz of C once.
#include "spam_internal.hpp"

namespace spam {

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void ewise_count_kernel(const uint64_t* __restrict__ a_rp, const uint32_t* __restrict__ a_col,
                                                         const uint64_t* __restrict__ b_rp, const uint32_t* __restrict__ b_col,
                                                         uint32_t rows, uint32_t* __restrict__ row_nnz) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t r = wave; r < rows; r += nwaves) {
    const uint64_t as = a_rp[r], bs = b_rp[r];
    const uint32_t la = (uint32_t)(a_rp[r + 1] - as), lb = (uint32_t)(b_rp[r + 1] - bs);
    uint32_t common = 0;
    for (uint32_t i = lane; i < la; i += 64) {
      const uint32_t c = a_col[as + i];
      const uint32_t p = lower_bound_u32(b_col + bs, lb, c);
      common += (p < lb && b_col[bs + p] == c);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) common += __shfl_xor(common, d, 64);
    if (lane == 0) row_nnz[r] = la + lb - common;
  }
}

template <class T>
__global__ __launch_bounds__(256) void ewise_fill_kernel(const uint64_t* __restrict__ a_rp, const uint32_t* __restrict__ a_col,
                                                        const T* __restrict__ a_val, const uint64_t* __restrict__ b_rp,
                                                        const uint32_t* __restrict__ b_col, const T* __restrict__ b_val,
                                                        uint32_t rows, int subtract, const uint64_t* __restrict__ c_rp,
                                                        uint32_t* __restrict__ c_col, T* __restrict__ c_val) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t r = wave; r < rows; r += nwaves) {
    const uint64_t as = a_rp[r], bs = b_rp[r], cs = c_rp[r];
    const uint32_t la = (uint32_t)(a_rp[r + 1] - as), lb = (uint32_t)(b_rp[r + 1] - bs);
    // A's entries: position = i + (#B < c) - (#matched A entries before i)
    uint32_t carry = 0;
    for (uint32_t base = 0; base < la; base += 64) {
      const uint32_t i = base + lane;
      uint32_t c = 0, p = 0, m = 0;
      T v = T(0);
      if (i < la) {
        c = a_col[as + i];
        v = a_val[as + i];
        p = lower_bound_u32(b_col + bs, lb, c);
        m = (p < lb && b_col[bs + p] == c);
        if (m) v = subtract ? v - b_val[bs + p] : v + b_val[bs + p];  // f(t1, t2) (lib.rs:114)
      }
      const uint64_t bal = __ballot(m != 0);
      if (i < la) {
        const uint32_t before = carry + __popcll(bal & ((1ull << lane) - 1));
        c_col[cs + i + p - before] = c;
        c_val[cs + i + p - before] = v;   // f(t, 0) = t for entries only in A (lib.rs:115)
      }
      carry += __popcll(bal);
    }
    // B's entries absent from A: position = j + (#A < c) - (#matched B entries before j)
    carry = 0;
    for (uint32_t base = 0; base < lb; base += 64) {
      const uint32_t j = base + lane;
      uint32_t c = 0, p = 0, m = 0;
      if (j < lb) {
        c = b_col[bs + j];
        p = lower_bound_u32(a_col + as, la, c);
        m = (p < la && a_col[as + p] == c);
      }
      const uint64_t bal = __ballot(m != 0);
      if (j < lb && !m) {
        const uint32_t before = carry + __popcll(bal & ((1ull << lane) - 1));
        const T t = b_val[bs + j];
        c_col[cs + j + p - before] = c;
        c_val[cs + j + p - before] = subtract ? T(0) - t : T(0) + t;  // f(0, t) (lib.rs:116)
      }
      carry += __popcll(bal);
    }
  }
}

}  // namespace spam

using namespace spam;

extern "C" int spam_csr_sort_rows(spam_csr* m);

static int private_sorted_copy(const spam_csr* m, spam_csr** out) {
  // upload-equivalent deep copy on the device, then sort its rows
  spam_csr* c = new (std::nothrow) spam_csr();
  if (!c) return set_error(SPAM_E_OOM, "host allocation failed");
  c->rows = m->rows, c->cols = m->cols, c->nnz = m->nnz, c->dtype = m->dtype, c->is_sorted = m->is_sorted;
  c->row_ptr = m->row_ptr;  // borrowed; col_idx / vals are replaced by private copies below
  DevBuf ci, va;
  hipStream_t s = stream();
  const size_t vs = dtype_size(m->dtype);
  if (int e = ci.alloc(m->nnz * 4)) { delete c; return e; }
  if (int e = va.alloc(m->nnz * vs)) { delete c; return e; }
  if (int e = wait_values(m, s)) { delete c; return e; }
  hipError_t h1 = hipMemcpyAsync(ci.p, m->col_idx, m->nnz * 4, hipMemcpyDeviceToDevice, s);
  hipError_t h2 = hipMemcpyAsync(va.p, m->vals, m->nnz * vs, hipMemcpyDeviceToDevice, s);
  if (h1 != hipSuccess || h2 != hipSuccess) { delete c; return set_error(SPAM_E_HIP, "device copy failed"); }
  c->col_idx = (uint32_t*)ci.release();
  c->vals = va.release();
  c->owns_col_val = true;
  if (int e = spam_csr_sort_rows(c)) { spam_csr_free(c); return e; }
  *out = c;
  return SPAM_OK;
}

template <class T>
static int add_typed(const spam_csr* a, const spam_csr* b, int subtract, spam_csr** out) {
  hipStream_t s = stream();
  SPAM_TRY(wait_values(a, s));
  SPAM_TRY(wait_values(b, s));
  const uint32_t rows = (uint32_t)a->rows;
  DevBuf rn, crp, tmp, ccol, cval;
  SPAM_TRY(rn.alloc((uint64_t)rows * 4));
  SPAM_TRY(crp.alloc(((uint64_t)rows + 1) * 8));
  SPAM_TRY(tmp.alloc(scan_tmp_bytes(rows)));
  const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)rows + 3) / 4, 32768);
  ewise_count_kernel<<<grid, 256, 0, s>>>(a->row_ptr, a->col_idx, b->row_ptr, b->col_idx, rows, rn.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  SPAM_TRY(scan_u32_to_u64(rn.as<uint32_t>(), rows, crp.as<uint64_t>(), tmp.p, s));
  uint64_t nnz = 0;
  SPAM_HIP_TRY(hipMemcpyAsync(&nnz, crp.as<uint64_t>() + rows, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  SPAM_TRY(ccol.alloc(nnz * 4));
  SPAM_TRY(cval.alloc(nnz * sizeof(T)));
  ewise_fill_kernel<T><<<grid, 256, 0, s>>>(a->row_ptr, a->col_idx, (const T*)a->vals, b->row_ptr, b->col_idx,
                                           (const T*)b->vals, rows, subtract, crp.as<uint64_t>(), ccol.as<uint32_t>(),
                                           cval.as<T>());
  SPAM_CHECK_LAUNCH();
  spam_csr* c = new (std::nothrow) spam_csr();
  if (!c) return set_error(SPAM_E_OOM, "host allocation failed");
  c->rows = a->rows, c->cols = a->cols, c->nnz = nnz, c->dtype = a->dtype, c->is_sorted = 1;
  c->row_ptr = (uint64_t*)crp.release();
  c->col_idx = (uint32_t*)ccol.release();
  c->vals = cval.release();
  c->owns_row_ptr = c->owns_col_val = true;
  if (nnz < (1ull << 32)) {
    DevBuf r32;
    if (int e = r32.alloc((c->rows + 1) * 4)) { spam_csr_free(c); return e; }
    c->row_ptr32 = (uint32_t*)r32.release();
    c->owns_row_ptr32 = true;
    if (int e = narrow_row_ptr(c->row_ptr, c->rows + 1, c->row_ptr32, s)) { spam_csr_free(c); return e; }
  }
  *out = c;
  return SPAM_OK;
}

extern "C" int spam_csr_add(const spam_csr* a, const spam_csr* b, int subtract, spam_csr** out) {
  clear_error();
  if (!a || !b || !out) return set_error(SPAM_E_INVALID, "NULL argument");
  *out = nullptr;
  if (a->rows != b->rows || a->cols != b->cols)
    return set_error(SPAM_E_INVALID, "matrices must have identical dimensions (assert_eq!, spam_csr/src/lib.rs:87-91)");
  if (a->dtype != b->dtype) return set_error(SPAM_E_DTYPE, "A and B have different value types");
  spam_csr *as = nullptr, *bs = nullptr;
  if (!a->is_sorted) SPAM_TRY(private_sorted_copy(a, &as));
  if (!b->is_sorted)
    if (int e = private_sorted_copy(b, &bs)) {
      if (as) spam_csr_free(as);
      return e;
    }
  const spam_csr* aa = as ? as : a;
  const spam_csr* bb = bs ? bs : b;
  int e = SPAM_DISPATCH_DTYPE(a->dtype, add_typed<double>(aa, bb, subtract, out), add_typed<float>(aa, bb, subtract, out),
                              add_typed<i64w>(aa, bb, subtract, out));
  if (e == SPAM_OK) (void)hipStreamSynchronize(stream());
  if (as) spam_csr_free(as);
  if (bs) spam_csr_free(bs);
  return e;
}
