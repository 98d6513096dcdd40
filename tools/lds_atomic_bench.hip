// Microbenchmark: LDS atomic throughput on MI355X (per CU), by op and address pattern.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: ds_add_f64 no return; 1: ds_add_f32 no return; 2: ds_add_u32 no return; 3: cmpst_rtn (CAS, wait each);
// 4: plain ds_write_b64 (no atomic) ; 5: ds_add_rtn_u32 with wait; 6: ds_read_b64 + add + ds_write_b64 (non-atomic RMW)
// PAT 0: random slot in [0, N); 1: lane-contiguous (conflict-free)
template <int MODE, int PAT>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, uint32_t nslots) {
  __shared__ double acc[16384];
  uint32_t* a32 = (uint32_t*)acc;
  float* f32 = (float*)acc;
  for (int i = threadIdx.x; i < 16384; i += 1024) acc[i] = 0;
  __syncthreads();
  uint32_t h = mix32(threadIdx.x + blockIdx.x * 1024u);
  uint32_t sink = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = mix32(h + u);
      uint32_t s = PAT == 0 ? (h % nslots) : ((threadIdx.x + u * 1024 + it * 64) % nslots);
      if (MODE == 0) atomicAdd(&acc[s], 1.0);
      else if (MODE == 1) atomicAdd(&f32[s], 1.0f);
      else if (MODE == 2) atomicAdd(&a32[s], 1u);
      else if (MODE == 3) sink += atomicCAS(&a32[s], 0xFFFFFFFFu, h);
      else if (MODE == 4) acc[s] = (double)h;
      else if (MODE == 5) sink += atomicAdd(&a32[s], 1u);
      else if (MODE == 6) { double v = acc[s]; acc[s] = v + 1.0; }
      else if (MODE == 7) {  // f32 add as a CAS loop on the bit pattern
        uint32_t old = a32[s], assumed;
        do { assumed = old; old = atomicCAS(&a32[s], assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f)); } while (old != assumed);
      } else if (MODE == 8) {  // f32 add with the returning form
        sink += __float_as_uint(atomicAdd(&f32[s], 1.0f));
      } else { sink += __float_as_uint(__hip_atomic_fetch_add(&f32[s], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
    }
  }
  __syncthreads();
  if (sink == 0x12345) out[0] = sink;
  if (threadIdx.x == 0) out[blockIdx.x + 1] = a32[5];
}
template <int MODE, int PAT>
int run(const char* name, uint32_t nslots) {
  uint32_t* out; CK(hipMalloc(&out, 4096 * 4));
  const int iters = 2000, blocks = 256;
  k<MODE, PAT><<<blocks, 1024>>>(out, 10, nslots);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  k<MODE, PAT><<<blocks, 1024>>>(out, iters, nslots);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double ops = (double)blocks * 1024 * iters * 8;
  printf("%-44s slots %6u: %8.3f ms  %7.1f Gops/s chip  %.3f lane-ops/clk/CU (2.1 GHz)\n", name, nslots, ms, ops / ms / 1e6, ops / 256 / (ms * 1e-3 * 2.1e9));
  CK(hipFree(out));
  return 0;
}
int main() {
  run<0, 0>("ds_add_f64 noret, random", 16384);
  run<0, 0>("ds_add_f64 noret, random", 512);
  run<0, 1>("ds_add_f64 noret, lane-contiguous", 16384);
  run<1, 0>("ds_add_f32 noret, random", 16384);
  run<1, 1>("ds_add_f32 noret, lane-contiguous", 16384);
  run<2, 0>("ds_add_u32 noret, random", 16384);
  run<2, 1>("ds_add_u32 noret, lane-contiguous", 16384);
  run<5, 0>("ds_add_rtn_u32 (waited), random", 16384);
  run<3, 0>("ds_cmpst_rtn_b32 (waited), random", 16384);
  run<3, 1>("ds_cmpst_rtn_b32 (waited), lane-contiguous", 16384);
  run<4, 0>("ds_write_b64 plain, random", 16384);
  run<6, 0>("ds_read_b64+add+ds_write_b64 (non-atomic), random", 16384);
  run<7, 0>("f32 add via CAS loop, random", 16384);
  run<8, 0>("ds_add_rtn_f32 (waited), random", 16384);
  run<9, 0>("__hip_atomic_fetch_add f32 workgroup scope", 16384);
  return 0;
}
