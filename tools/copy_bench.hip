// Microbenchmark: f4v device copy variants on MI355X (which one is the streaming ceiling bench.py should quote).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const f4v* __restrict__ in, f4v* __restrict__ out, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) out[i] = in[i];
}
template <int U, bool NT>
int run(const char* name, uint64_t bytes, unsigned grid) {
  f4v *a, *b; const uint64_t n = bytes / 16;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<U, NT><<<grid, 256>>>(a, b, n);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) k<U, NT><<<grid, 256>>>(a, b, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s %5.1f GiB grid %6u: %8.1f GB/s (read + write)\n", name, bytes / 1073741824.0, grid, 2.0 * bytes * 10 / (ms * 1e-3) / 1e9);
  CK(hipFree(a)); CK(hipFree(b));
  return 0;
}
int main() {
  for (uint64_t gib : {1ull, 4ull}) {
    const uint64_t bytes = gib << 30;
    for (unsigned grid : {2048u, 8192u, 16384u, 65536u}) {
      run<1, false>("plain", bytes, grid);
      run<4, false>("unroll 4", bytes, grid);
      run<4, true>("unroll 4 nontemporal", bytes, grid);
      run<8, false>("unroll 8", bytes, grid);
    }
  }
  return 0;
}
