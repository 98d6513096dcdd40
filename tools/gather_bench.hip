// Microbenchmark: random-gather throughput on MI355X as a function of table footprint, element size and
// cache policy.  Diagnostic only (not product, not tests).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

// idx[] precomputed (coalesced u32 stream, like ja); gather table[idx]
template <class T, int MODE, int NPT>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ idx, const T* __restrict__ table, T* __restrict__ out, uint64_t n) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint64_t base = (uint64_t)blockIdx.x * 256 * NPT;
  T acc = 0;
  u32x4 c[NPT / 4];
#pragma unroll
  for (int i = 0; i < NPT / 4; ++i) c[i] = __builtin_nontemporal_load((const u32x4*)(idx + base + (uint64_t)(i * 256 + threadIdx.x) * 4));
#pragma unroll
  for (int i = 0; i < NPT / 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T* p = table + c[i][j];
      T v;
      if (MODE == 0) v = *p;
      else if (MODE == 1) v = __builtin_nontemporal_load(p);
      else v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += v;
    }
  out[(uint64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void fill_idx(uint32_t* idx, uint64_t n, uint64_t tsize, int skew_bits, uint64_t seed) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t h = mix(i + seed);
  uint32_t v;
  if (skew_bits == 0) v = (uint32_t)(h % tsize);
  else {  // R-MAT-like: each bit is 1 with prob 0.24
    v = 0; uint64_t hh = h;
    for (int b = 0; b < skew_bits; ++b) { if ((b & 3) == 0) hh = mix(h + b); uint32_t u = (hh >> (16 * (b & 3))) & 0xFFFF; v = (v << 1) | (u >= 49807u); }
    if (v >= tsize) v = (uint32_t)(h % tsize);
  }
  idx[i] = v;
}

static int g_alloc_mode = 0;  // 0 hipMalloc, 1 uncached, 2 fine-grained, 3 host pinned
template <class T, int MODE>
int run(const char* name, uint64_t tsize, int skew_bits) {
  const uint64_t n = 1ull << 27;  // 134M gathers
  constexpr int NPT = 16;
  uint32_t* idx; T* table; T* out;
  CK(hipMalloc(&idx, n * 4)); if (g_alloc_mode == 0) CK(hipMalloc(&table, tsize * sizeof(T)));
  else if (g_alloc_mode == 1) CK(hipExtMallocWithFlags((void**)&table, tsize * sizeof(T), hipDeviceMallocUncached));
  else if (g_alloc_mode == 2) CK(hipExtMallocWithFlags((void**)&table, tsize * sizeof(T), hipDeviceMallocFinegrained));
  else CK(hipHostMalloc((void**)&table, tsize * sizeof(T), hipHostMallocDefault)); CK(hipMalloc(&out, (n / NPT) * sizeof(T)));
  CK(hipMemset(table, 0, tsize * sizeof(T)));
  fill_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, tsize, skew_bits, 12345);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  unsigned grid = (unsigned)(n / (256 * NPT));
  for (int w = 0; w < 2; ++w) gather_kernel<T, MODE, NPT><<<grid, 256>>>(idx, table, out, n);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) gather_kernel<T, MODE, NPT><<<grid, 256>>>(idx, table, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  printf("%-28s elem %zuB table %8.1f MB skew %2d: %.3f ms  %.1f Ggather/s  (idx stream alone would be %.3f ms at 6 TB/s)\n", name, sizeof(T),
         tsize * sizeof(T) / 1e6, skew_bits, ms, n / ms / 1e6, n * 4 / 6e9);
  CK(hipFree(idx)); if (g_alloc_mode == 3) CK(hipHostFree(table)); else CK(hipFree(table)); CK(hipFree(out));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1) {
    for (int m : {1, 2}) {
      g_alloc_mode = m;
      printf("--- alloc mode %d (1 uncached, 2 fine-grained)\n", m);
      for (uint64_t mb : {3ull, 80ull, 1000ull}) run<double, 0>("f64 plain uniform", mb * 1000000 / 8, 0);
      run<double, 0>("f64 plain rmat-skew 24b", 10000000, 24);
      run<double, 1>("f64 nt rmat-skew 24b", 10000000, 24);
      run<float, 0>("f32 plain uniform 80MB", 20000000, 0);
    }
    return 0;
  }
  for (uint64_t mb : {1ull, 3ull, 8ull, 32ull, 80ull, 200ull, 1000ull}) {
    run<double, 0>("f64 plain uniform", mb * 1000000 / 8, 0);
  }
  for (uint64_t mb : {1ull, 8ull, 80ull}) run<float, 0>("f32 plain uniform", mb * 1000000 / 4, 0);
  run<double, 1>("f64 nt uniform", 80ull * 1000000 / 8, 0);
  run<double, 2>("f64 sc1 uniform", 80ull * 1000000 / 8, 0);
  run<double, 0>("f64 plain rmat-skew 24b", 10000000, 24);
  run<double, 1>("f64 nt rmat-skew 24b", 10000000, 24);
  run<double, 2>("f64 sc1 rmat-skew 24b", 10000000, 24);
  run<float, 0>("f32 plain rmat-skew 24b", 10000000, 24);
  return 0;
}
