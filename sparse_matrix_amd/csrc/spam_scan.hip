// spam_scan.hip — device prefix sum: out = [0, v0, v0+v1, ...]  (the shape of checked_inclusive_scan,
// spam_csr/src/lib.rs:267-274, which the reference runs serially on the CPU twice per product,
// mul_hash.rs:51,117).  Three launches: per-block sums, one block scanning the block sums, per-block
// scan + offset.  HBM-bound: reads n*sizeof(In) twice, writes (n+1)*8 once.
// Overflow (the reference panics on checked_add, lib.rs:270): the sums are u64 and cannot wrap for either use —
// u32 inputs (entries per row of C, flags) over fewer than 2^32 elements stay below 2^64; the u64 input (per-row
// partition costs) is admitted by spam_rows_to_parts* only when the total F, itself overflow-checked while it is
// counted (spgemm_flops_kernel), is below 2^63.
#include "spam_internal.hpp"

namespace spam {

constexpr int SCAN_TPB = 256;
constexpr int SCAN_IPT = 8;                       // items per thread
constexpr int SCAN_BLOCK = SCAN_TPB * SCAN_IPT;   // 2048 items per block

__device__ inline uint64_t wave_incl_scan_u64(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns exclusive prefix, total in *total
__device__ inline uint64_t block_excl_scan_u64(uint64_t v, uint64_t* total) {
  __shared__ uint64_t wsum[SCAN_TPB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t inc = wave_incl_scan_u64(v, lane);
  __syncthreads();  // protect wsum reuse across calls
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint64_t off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_TPB / 64; ++i) {
    if (i < w) off += wsum[i];
    tot += wsum[i];
  }
  *total = tot;
  return off + inc - v;
}

template <class In>
__global__ __launch_bounds__(SCAN_TPB) void scan_block_sums_kernel(const In* __restrict__ in, uint64_t n,
                                                                  uint64_t* __restrict__ block_sums) {
  uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_IPT; ++i) {
    uint64_t idx = base + (uint64_t)i * SCAN_TPB + threadIdx.x;
    if (idx < n) s += (uint64_t)in[idx];
  }
  uint64_t tot;
  block_excl_scan_u64(s, &tot);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// one block: exclusive scan of block_sums in place (nb may exceed the block: loop with a running carry)
__global__ __launch_bounds__(SCAN_TPB) void scan_sums_kernel(uint64_t* __restrict__ block_sums, uint64_t nb) {
  uint64_t carry = 0;
  for (uint64_t base = 0; base < nb; base += SCAN_TPB) {
    uint64_t idx = base + threadIdx.x;
    uint64_t v = idx < nb ? block_sums[idx] : 0;
    uint64_t tot;
    uint64_t ex = block_excl_scan_u64(v, &tot);
    if (idx < nb) block_sums[idx] = carry + ex;
    carry += tot;
  }
}

template <class In>
__global__ __launch_bounds__(SCAN_TPB) void scan_apply_kernel(const In* __restrict__ in, uint64_t n,
                                                             const uint64_t* __restrict__ block_offs,
                                                             uint64_t* __restrict__ out) {
  // thread t owns SCAN_IPT consecutive items so the per-thread serial part is a plain running sum
  uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_IPT;
  uint64_t v[SCAN_IPT];
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_IPT; ++i) {
    v[i] = (base + i < n) ? (uint64_t)in[base + i] : 0;
    s += v[i];
  }
  uint64_t tot;
  uint64_t run = block_excl_scan_u64(s, &tot) + block_offs[blockIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
#pragma unroll
  for (int i = 0; i < SCAN_IPT; ++i) {
    run += v[i];
    if (base + i < n) out[base + i + 1] = run;
  }
}

size_t scan_tmp_bytes(uint64_t n) { return ((n + SCAN_BLOCK - 1) / SCAN_BLOCK + 1) * sizeof(uint64_t); }

template <class In>
static int scan_impl(const In* d_in, uint64_t n, uint64_t* d_out, void* d_tmp, hipStream_t s) {
  if (n == 0) {
    SPAM_HIP_TRY(hipMemsetAsync(d_out, 0, 8, s));
    return SPAM_OK;
  }
  uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  uint64_t* sums = (uint64_t*)d_tmp;
  scan_block_sums_kernel<In><<<(unsigned)nb, SCAN_TPB, 0, s>>>(d_in, n, sums);
  SPAM_CHECK_LAUNCH();
  scan_sums_kernel<<<1, SCAN_TPB, 0, s>>>(sums, nb);
  SPAM_CHECK_LAUNCH();
  scan_apply_kernel<In><<<(unsigned)nb, SCAN_TPB, 0, s>>>(d_in, n, sums, d_out);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

int scan_u32_to_u64(const uint32_t* d_in, uint64_t n, uint64_t* d_out, void* d_tmp, hipStream_t s) {
  return scan_impl<uint32_t>(d_in, n, d_out, d_tmp, s);
}
int scan_u64_to_u64(const uint64_t* d_in, uint64_t n, uint64_t* d_out, void* d_tmp, hipStream_t s) {
  return scan_impl<uint64_t>(d_in, n, d_out, d_tmp, s);
}

}  // namespace spam
