// Diagnostic: does a gather instruction with fewer ACTIVE lanes cost the vector-memory pipeline less?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ta_lanes_bench tools/ta_lanes_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// every `stride`-th lane gathers (active = 64 / stride lanes); addresses random in [0, T) or sorted with small gaps
__global__ __launch_bounds__(512) void k(const double* __restrict__ tab, uint32_t T, int stride, int sorted, int iters, double* out) {
  const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const bool act = (lane % stride) == 0;
  double s = 0;
  for (int it = 0; it < iters; it += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t base = mix((gw * 131071u + it + u) * 2246822519u);
      const uint32_t idx = (sorted ? base + lane * 3u : mix(base + lane * 7919u)) & (T - 1);
      v[u] = 0;
      if (act) v[u] = tab[idx];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  if (s == 1.2345) out[0] = s;
}
int main() {
  const uint32_t T = 1u << 18;  // 2 MB: L2-resident
  double *tab, *out;
  hipMalloc(&tab, T * 8); hipMalloc(&out, 8);
  hipMemset(tab, 0, T * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2048, grid = 512;
  for (int sorted : {0, 1})
    for (int stride : {1, 2, 4, 8, 16}) {
      k<<<grid, 512>>>(tab, T, stride, sorted, 64, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<grid, 512>>>(tab, T, stride, sorted, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr = (double)grid * 8 * iters;  // wave instructions
      printf("%-7s addresses, %2d active lanes: %7.3f ms  %7.2f G wave-gathers/s  %6.1f cycles per wave-gather and CU (2.4 GHz)\n",
             sorted ? "sorted" : "random", 64 / stride, ms, instr / ms / 1e6, ms * 1e-3 * 2.4e9 / (instr / 256));
    }
  return 0;
}
