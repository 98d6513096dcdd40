// Diagnostic: rate of 8-byte gathers served by the CU's vector L1 / by LDS, by address pattern within a wave instruction.
//   hipcc --offload-arch=gfx950 -O3 -o tools/l1_gather_bench tools/l1_gather_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// mode 0: random in [0, T); mode 1: base(it) + lane * g  (g = 0 broadcast, 1 coalesced, 16 = one 128-B line per lane)
// mode 2: sorted random: base(it) + lane-prefix of small random gaps with mean g
template <int LDS>
__global__ __launch_bounds__(1024) void gather_kernel(const double* __restrict__ tab, uint32_t T, int mode, int g, int iters,
                                                      double* out) {
  extern __shared__ double lt[];
  if (LDS) {
    for (uint32_t i = threadIdx.x; i < T; i += blockDim.x) lt[i] = tab[i];
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  double s = 0;
  uint32_t gap = 0;
  if (mode == 2) {  // per-lane offset = prefix sum of random gaps in [0, 2g]
    uint32_t mygap = g ? mix(threadIdx.x * 2654435761u + 17) % (2 * g + 1) : 0, pre = 0;
    for (int l = 0; l < 64; ++l) {
      uint32_t v = __shfl(mygap, l);
      if ((uint32_t)l < lane) pre += v;
    }
    gap = pre;
  }
  for (int it = 0; it < iters; it += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      uint32_t base = mix((gw * 131071u + it + u) * 2246822519u);
      uint32_t idx;
      if (mode == 0) idx = mix(base + lane * 7919u);
      else if (mode == 1) idx = base + lane * g;
      else idx = base + gap;
      idx &= (T - 1);
      v[u] = LDS ? lt[idx] : tab[idx];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  if (s == 1.2345) out[0] = s;
}

int main() {
  const uint32_t TMAX = 1u << 22;
  double *tab, *out;
  hipMalloc(&tab, TMAX * 8);
  hipMalloc(&out, 8);
  std::vector<double> h(TMAX, 1.0);
  hipMemcpy(tab, h.data(), TMAX * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4096, grid = 256;
  struct C { int lds; uint32_t T; int mode, g; const char* name; };
  std::vector<C> cs;
  for (uint32_t T : {2048u, 4096u, 16384u, 1u << 18}) {
    cs.push_back({0, T, 0, 0, "random"});
    cs.push_back({0, T, 1, 0, "broadcast"});
    cs.push_back({0, T, 1, 1, "coalesced (8 B stride)"});
    cs.push_back({0, T, 1, 2, "16 B stride"});
    cs.push_back({0, T, 1, 4, "32 B stride"});
    cs.push_back({0, T, 1, 16, "128 B stride (64 lines)"});
    cs.push_back({0, T, 2, 1, "sorted, mean gap 1"});
    cs.push_back({0, T, 2, 4, "sorted, mean gap 4"});
    cs.push_back({0, T, 2, 16, "sorted, mean gap 16"});
  }
  cs.push_back({1, 2048, 0, 0, "LDS random"});
  cs.push_back({1, 16384, 0, 0, "LDS random"});
  cs.push_back({1, 16384, 2, 4, "LDS sorted, mean gap 4"});
  cs.push_back({1, 16384, 1, 1, "LDS coalesced"});
  for (auto& c : cs) {
    size_t lds = c.lds ? c.T * 8 : 0;
    auto k = c.lds ? gather_kernel<1> : gather_kernel<0>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<<<grid, 1024, lds>>>(tab, c.T, c.mode, c.g, 64, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<grid, 1024, lds>>>(tab, c.T, c.mode, c.g, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double n = (double)grid * 1024 * iters;
    printf("%-4s table %8.1f KB  %-28s: %8.3f ms  %8.1f Ggather/s chip  %6.2f lanes/clk/CU (2.4 GHz)\n", c.lds ? "LDS" : "L1/2",
           c.T * 8 / 1024.0, c.name, ms, n / ms / 1e6, n / ms / 1e6 / 256 / 2.4);
  }
  return 0;
}
