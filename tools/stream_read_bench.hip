// Diagnostic: what read rate do different streaming structures reach on MI355X?  (Which one should the row-group SpMV copy.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_read_bench tools/stream_read_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// V1: grid-stride 16-byte loads, U loads in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void v1(const u32x4* __restrict__ in, uint64_t n, uint32_t* out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// V2: persistent workgroups take blocks of `bchunks` 3 KB chunks by ticket; the chunks of a block are dealt to the
// wavefronts round-robin (RR) or in contiguous runs; a wavefront keeps K chunks (3 x 16 B per lane each) in flight.
// LDSOPS: 4 LDS atomic adds per chunk and lane, as the SpMV kernel does.
template <int NT_, int K, bool RR, int LDSOPS>
__global__ __launch_bounds__(NT_) void v2(const unsigned char* __restrict__ in, uint32_t nblocks, uint32_t bchunks, uint32_t* ticket,
                                          uint32_t* out) {
  __shared__ uint32_t s_blk;
  __shared__ double acc[8192];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int nw = NT_ / 64;
  uint32_t sum = 0;
  if (LDSOPS) for (int i = threadIdx.x; i < 8192; i += NT_) acc[i] = 0;
  for (;;) {
    if (threadIdx.x == 0) s_blk = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t b = s_blk;
    __syncthreads();
    if (b >= nblocks) break;
    const unsigned char* p0 = in + (uint64_t)b * bchunks * 3072;
    uint32_t c, cend, cstep;
    if (RR) c = wave, cend = bchunks, cstep = nw;
    else { const uint32_t per = (bchunks + nw - 1) / nw; c = wave * per; cend = min(bchunks, c + per); cstep = 1; }
    for (; c + (K - 1) * cstep < cend; c += K * cstep) {
      u32x4 w[K], v0[K], v1[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const unsigned char* q = p0 + (uint64_t)(c + k * cstep) * 3072;
        w[k] = __builtin_nontemporal_load((const u32x4*)(q + 16 * lane));
        v0[k] = __builtin_nontemporal_load((const u32x4*)(q + 1024 + 16 * lane));
        v1[k] = __builtin_nontemporal_load((const u32x4*)(q + 2048 + 16 * lane));
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (LDSOPS == 1) {        // 4 atomics per lane and chunk, accumulator from the entry word
          atomicAdd(&acc[w[k].x & 8191], __hiloint2double(v0[k].y, v0[k].x));
          atomicAdd(&acc[w[k].y & 8191], __hiloint2double(v0[k].w, v0[k].z));
          atomicAdd(&acc[w[k].z & 8191], __hiloint2double(v1[k].y, v1[k].x));
          atomicAdd(&acc[w[k].w & 8191], __hiloint2double(v1[k].w, v1[k].z));
        } else if (LDSOPS == 2) { // 2 atomics per lane and chunk
          atomicAdd(&acc[w[k].x & 8191], __hiloint2double(v0[k].y, v0[k].x) + __hiloint2double(v0[k].w, v0[k].z));
          atomicAdd(&acc[w[k].z & 8191], __hiloint2double(v1[k].y, v1[k].x) + __hiloint2double(v1[k].w, v1[k].z));
        } else if (LDSOPS == 3) { // 4 atomics, lane-contiguous accumulators (no bank conflicts)
          atomicAdd(&acc[(lane + 64 * 0 + (w[k].x & 7168)) & 8191], __hiloint2double(v0[k].y, v0[k].x));
          atomicAdd(&acc[(lane + 64 * 1 + (w[k].x & 7168)) & 8191], __hiloint2double(v0[k].w, v0[k].z));
          atomicAdd(&acc[(lane + 64 * 2 + (w[k].x & 7168)) & 8191], __hiloint2double(v1[k].y, v1[k].x));
          atomicAdd(&acc[(lane + 64 * 3 + (w[k].x & 7168)) & 8191], __hiloint2double(v1[k].w, v1[k].z));
        } else {
          sum += w[k].x ^ v0[k].y ^ v1[k].z;
        }
      }
    }
    __syncthreads();
  }
  if (LDSOPS) { __syncthreads(); if (acc[threadIdx.x] == 1.2345) out[1] = 1; }
  if (sum == 0x12345678u) out[0] = sum;
}

__global__ void fill(uint32_t* p, uint64_t n, int skew) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
    // skew: accumulators of a hub-heavy block (a quarter of the entries in 64 accumulators)
    p[i] = (skew && (x & 0x30000u) == 0) ? (x & 63u) : x;
  }
}
int main(int argc, char** argv) {
  const int skew = argc > 1 ? atoi(argv[1]) : 0;
  const uint64_t bytes = 1920ull << 20;  // multiple of 3072 * many
  unsigned char* buf; uint32_t *out, *ticket;
  hipMalloc(&buf, bytes); hipMalloc(&out, 64); hipMalloc(&ticket, 64);
  fill<<<8192, 256>>>((uint32_t*)buf, bytes / 4, skew);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-78s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  const uint64_t n16 = bytes / 16;
  time("grid-stride 16 B loads, U=4 nt, grid 8192", [&] { v1<4, true><<<8192, 256>>>((const u32x4*)buf, bytes / 16, out); });
  const uint64_t nchunks = bytes / 3072;
  for (uint32_t bch : {400u}) {
    const uint32_t nb = (uint32_t)(nchunks / bch);
    char nm[160];
#define RUN2(NT_, K_, RR_, L_, GRID)                                                                                         \
  snprintf(nm, 160, "tickets: %4u-chunk blocks, %4d thr x %d WG/CU, K=%d, %s, %s", bch, NT_, GRID / 256, K_, RR_ ? "round-robin" : "contiguous", \
           L_ == 0 ? "no LDS" : L_ == 1 ? "4 LDS atomics" : L_ == 2 ? "2 LDS atomics" : "4 LDS atomics, lane-contiguous");                                                                         \
  time(nm, [&] { hipMemsetAsync(ticket, 0, 4); v2<NT_, K_, RR_, L_><<<GRID, NT_>>>(buf, nb, bch, ticket, out); });
    RUN2(512, 4, true, 0, 512)
    RUN2(512, 4, true, 1, 512)
    RUN2(512, 4, true, 2, 512)
    RUN2(512, 4, true, 3, 512)
    RUN2(1024, 2, true, 1, 512)
    RUN2(256, 4, true, 1, 1024)
  }
  return 0;
}
