// Microbenchmark: do the eight per-XCD L2s of MI355X act as ONE 8 x S cache for random gathers when every
// workgroup only touches the table slice of the XCD it runs on?  Diagnostic only (not product, not tests).
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcd_gather_bench tools/xcd_gather_bench.hip
// Workgroups read HW_REG_XCC_ID, then pull 4096-gather tiles from that XCD's queue; tile indices of queue k
// are uniform in slice k of the table.  Compared with: the same tiles on arbitrary XCDs (every XCD touches
// the whole 8 x S table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__device__ inline uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v;
}

constexpr int NPT = 16, TPB = 256, TILE = NPT * TPB;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: slab = own XCD (queue per XCD).  MODE 1: slab = (pulled ticket) % 8, any XCD (control).
template <class T, int MODE>
__global__ __launch_bounds__(TPB) void gather_kernel(const uint32_t* __restrict__ idx, const T* __restrict__ table,
                                                     T* __restrict__ out, uint32_t tiles_per_slab,
                                                     uint32_t* __restrict__ heads, uint32_t* __restrict__ census) {
  __shared__ uint32_t s_t;
  const uint32_t xcc = xcc_id() & 7;
  if (threadIdx.x == 0 && census) atomicAdd(&census[xcc * 8 + (blockIdx.x & 7)], 1u);
  T acc = 0;
  for (;;) {
    if (threadIdx.x == 0) s_t = atomicAdd(&heads[MODE == 0 ? xcc : 8], 1u);
    __syncthreads();
    const uint32_t t = s_t;
    __syncthreads();
    uint32_t slab, tt;
    if (MODE == 0) {
      if (t >= tiles_per_slab) break;
      slab = xcc, tt = t;
    } else {
      if (t >= tiles_per_slab * 8) break;
      slab = t & 7, tt = t >> 3;
    }
    const uint64_t base = ((uint64_t)slab * tiles_per_slab + tt) * TILE;
    u32x4 c[NPT / 4];
#pragma unroll
    for (int i = 0; i < NPT / 4; ++i) c[i] = __builtin_nontemporal_load((const u32x4*)(idx + base + (uint64_t)(i * TPB + threadIdx.x) * 4));
#pragma unroll
    for (int i = 0; i < NPT / 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += table[c[i][j]];
  }
  out[(uint64_t)blockIdx.x * TPB + threadIdx.x] = acc;
}

// Closer to the SpMV tile: VAL streams 8 more bytes per gather (non-temporal), LDSW stages the products in LDS behind a
// barrier and lets every thread sum 16 consecutive ones afterwards.
template <class T, bool VAL, bool LDSW, int LDSK = 0>
__global__ __launch_bounds__(TPB) void gather2_kernel(const uint32_t* __restrict__ idx, const T* __restrict__ vals,
                                                      const T* __restrict__ table, T* __restrict__ out, uint32_t tiles_per_slab,
                                                      uint32_t* __restrict__ heads, uint32_t slice = 0) {
  __shared__ uint32_t s_t;
  __shared__ T prod[TILE + TILE / 32 + 1];
  __shared__ T ltab[LDSK ? LDSK : 1];
  const uint32_t xcc = xcc_id() & 7;
  if (LDSK) for (int j = threadIdx.x; j < LDSK; j += TPB) ltab[j] = table[(uint64_t)xcc * slice + j];
  typedef T VT __attribute__((ext_vector_type(16 / sizeof(T))));
  constexpr int VN = 16 / sizeof(T);
  T acc = 0;
  for (;;) {
    if (threadIdx.x == 0) s_t = atomicAdd(&heads[xcc * 64], 1u);
    __syncthreads();
    const uint32_t t = s_t;
    __syncthreads();
    if (t >= tiles_per_slab) break;
    const uint64_t base = ((uint64_t)xcc * tiles_per_slab + t) * TILE;
    u32x4 c[NPT / 4];
    VT v[NPT / 4][4 / VN];
#pragma unroll
    for (int i = 0; i < NPT / 4; ++i) {
      const uint64_t q = base + (uint64_t)(i * TPB + threadIdx.x) * 4;
      c[i] = __builtin_nontemporal_load((const u32x4*)(idx + q));
      if (VAL) {
#pragma unroll
        for (int j = 0; j < 4 / VN; ++j) v[i][j] = __builtin_nontemporal_load((const VT*)(vals + q) + j);
      }
    }
    T xv[NPT / 4][4];
#pragma unroll
    for (int i = 0; i < NPT / 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (LDSK) { const uint32_t loc = c[i][j] - xcc * slice; if (loc < (uint32_t)LDSK) xv[i][j] = ltab[loc]; else xv[i][j] = table[c[i][j]]; }
        else xv[i][j] = table[c[i][j]];
      }
#pragma unroll
    for (int i = 0; i < NPT / 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const T pr = VAL ? v[i][j / VN][j % VN] * xv[i][j] : xv[i][j];
        if (LDSW) { const int q = (i * TPB + threadIdx.x) * 4 + j; prod[q + (q >> 5)] = pr; }
        else acc += pr;
      }
    if (LDSW) {
      __syncthreads();
      for (int k = 0; k < NPT; ++k) { const int q = threadIdx.x * NPT + k; acc += prod[q + (q >> 5)]; }
    }
  }
  out[(uint64_t)blockIdx.x * TPB + threadIdx.x] = acc;
}

// skew = 0: uniform inside the slice; otherwise R-MAT-like (every bit of the slot is 1 with probability 0.24)
__global__ void fill_idx2(uint32_t* idx, uint64_t n, uint32_t tiles_per_slab, uint64_t slice, int skew_bits, uint64_t seed) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t slab = i / ((uint64_t)tiles_per_slab * TILE);
  uint64_t h = mix(i + seed), v;
  if (skew_bits == 0) v = h % slice;
  else if (skew_bits < 0) v = ((h >> 40) % 100 < (uint64_t)(-skew_bits)) ? (h % 2048) : (2048 + h % (slice - 2048));
  else {
    v = 0; uint64_t hh = h;
    for (int b = 0; b < skew_bits; ++b) { if ((b & 3) == 0) hh = mix(h + b); uint32_t u = (hh >> (16 * (b & 3))) & 0xFFFF; v = (v << 1) | (u >= 49807u); }
    v %= slice;
  }
  idx[i] = (uint32_t)(slab * slice + v);
}

template <class T, bool VAL, bool LDSW, int LDSK = 0>
int run2(double slice_mb, int wg_per_cu, int skew_bits) {
  const uint32_t tiles_per_slab = 4096;
  const uint64_t n = (uint64_t)tiles_per_slab * 8 * TILE;
  const uint64_t slice = (uint64_t)(slice_mb * 1048576 / sizeof(T));
  uint32_t *idx, *heads; T *table, *out, *vals;
  const unsigned grid = 256 * wg_per_cu;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&vals, n * sizeof(T))); CK(hipMalloc(&table, slice * 8 * sizeof(T)));
  CK(hipMalloc(&out, (uint64_t)grid * TPB * sizeof(T))); CK(hipMalloc(&heads, 8 * 256));
  CK(hipMemset(table, 0, slice * 8 * sizeof(T))); CK(hipMemset(vals, 0, n * sizeof(T)));
  fill_idx2<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, tiles_per_slab, slice, skew_bits, 999);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int reps = 5; float tot = 0;
  for (int r = -2; r < reps; ++r) {
    CK(hipMemsetAsync(heads, 0, 8 * 256, 0));
    CK(hipEventRecord(a));
    gather2_kernel<T, VAL, LDSW, LDSK><<<grid, TPB>>>(idx, vals, table, out, tiles_per_slab, heads, (uint32_t)slice);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float m; CK(hipEventElapsedTime(&m, a, b));
    if (r >= 0) tot += m;
  }
  const float ms = tot / reps;
  printf("tile-like: elem %zuB slice %.2f MB, %d WG/CU, skew bits %2d, val stream %d, LDS staging %d, LDS table %d slots: %.3f ms %.1f Ggather/s\n", sizeof(T), slice_mb,
         wg_per_cu, skew_bits, (int)VAL, (int)LDSW, LDSK, ms, n / ms / 1e6);
  CK(hipFree(idx)); CK(hipFree(vals)); CK(hipFree(table)); CK(hipFree(out)); CK(hipFree(heads));
  return 0;
}


// Fourth series: whole tiles are either LDS-kind (every index < LDSK, gathered from the workgroup's LDS copy with ds_read:
// no vector-memory gather at all) or table-kind; `pct` percent of the tiles are LDS-kind.  Workgroup shape and table size vary.
template <class T, int TPBX, int NPTX, int LDSK>
__global__ __launch_bounds__(TPBX) void gather3_kernel(const uint32_t* __restrict__ idx, const T* __restrict__ vals,
                                                       const T* __restrict__ table, T* __restrict__ out, uint32_t tiles_per_slab,
                                                       uint32_t* __restrict__ heads, uint32_t slice, uint32_t pct) {
  constexpr int TILEX = TPBX * NPTX;
  __shared__ uint32_t s_t;
  __shared__ T prod[TILEX + TILEX / 32 + 1];
  extern __shared__ __align__(16) unsigned char dyn[];
  T* ltab = (T*)dyn;
  const uint32_t xcc = xcc_id() & 7;
  for (int j = threadIdx.x; j < LDSK; j += TPBX) ltab[j] = table[(uint64_t)xcc * slice + j];
  typedef T VT __attribute__((ext_vector_type(16 / sizeof(T))));
  constexpr int VN = 16 / sizeof(T);
  T acc = 0;
  for (;;) {
    if (threadIdx.x == 0) s_t = atomicAdd(&heads[xcc * 64], 1u);
    __syncthreads();
    const uint32_t t = s_t;
    __syncthreads();
    if (t >= tiles_per_slab) break;
    const bool lds_tile = (t % 100u) < pct;
    const uint64_t base = ((uint64_t)xcc * tiles_per_slab + t) * TILEX;
    u32x4 c[NPTX / 4];
    VT v[NPTX / 4][4 / VN];
#pragma unroll
    for (int i = 0; i < NPTX / 4; ++i) {
      const uint64_t q = base + (uint64_t)(i * TPBX + threadIdx.x) * 4;
      c[i] = __builtin_nontemporal_load((const u32x4*)(idx + q));
#pragma unroll
      for (int j = 0; j < 4 / VN; ++j) v[i][j] = __builtin_nontemporal_load((const VT*)(vals + q) + j);
    }
    T xv[NPTX / 4][4];
    if (lds_tile) {
#pragma unroll
      for (int i = 0; i < NPTX / 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[i][j] = ltab[c[i][j] - xcc * slice];
    } else {
#pragma unroll
      for (int i = 0; i < NPTX / 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[i][j] = table[c[i][j]];
    }
#pragma unroll
    for (int i = 0; i < NPTX / 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int q = (i * TPBX + threadIdx.x) * 4 + j; prod[q + (q >> 5)] = v[i][j / VN][j % VN] * xv[i][j]; }
    __syncthreads();
    for (int k = 0; k < NPTX; ++k) { const int q = threadIdx.x * NPTX + k; acc += prod[q + (q >> 5)]; }
  }
  out[(uint64_t)blockIdx.x * TPBX + threadIdx.x] = acc;
}
__global__ void fill_idx3(uint32_t* idx, uint64_t n, uint32_t tiles_per_slab, uint32_t tile, uint64_t slice, uint32_t ldsk, uint32_t pct, uint64_t seed) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t slab = i / ((uint64_t)tiles_per_slab * tile);
  const uint32_t t = (uint32_t)((i / tile) % tiles_per_slab);
  const uint64_t h = mix(i + seed);
  const uint64_t v = (t % 100u) < pct ? h % ldsk : ldsk + h % (slice - ldsk);
  idx[i] = (uint32_t)(slab * slice + v);
}
template <class T, int TPBX, int NPTX, int LDSK>
int run3(double slice_mb, int wg_per_cu, uint32_t pct) {
  constexpr int TILEX = TPBX * NPTX;
  const uint32_t tiles_per_slab = (uint32_t)((1ull << 24) / TILEX);   // 16.8 M gathers per slab, 134 M in all
  const uint64_t n = (uint64_t)tiles_per_slab * 8 * TILEX;
  const uint64_t slice = (uint64_t)(slice_mb * 1048576 / sizeof(T));
  uint32_t *idx, *heads; T *table, *out, *vals;
  const unsigned grid = 256 * wg_per_cu;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&vals, n * sizeof(T))); CK(hipMalloc(&table, slice * 8 * sizeof(T)));
  CK(hipMalloc(&out, (uint64_t)grid * TPBX * sizeof(T))); CK(hipMalloc(&heads, 8 * 256));
  CK(hipMemset(table, 0, slice * 8 * sizeof(T))); CK(hipMemset(vals, 0, n * sizeof(T)));
  fill_idx3<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, tiles_per_slab, TILEX, slice, LDSK, pct, 999);
  CK(hipDeviceSynchronize());
  auto kern = gather3_kernel<T, TPBX, NPTX, LDSK>;
  const size_t dynb = (size_t)LDSK * sizeof(T);
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynb));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int reps = 5; float tot = 0;
  for (int r = -2; r < reps; ++r) {
    CK(hipMemsetAsync(heads, 0, 8 * 256, 0));
    CK(hipEventRecord(a));
    kern<<<grid, TPBX, dynb>>>(idx, vals, table, out, tiles_per_slab, heads, (uint32_t)slice, pct);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float m; CK(hipEventElapsedTime(&m, a, b));
    if (r >= 0) tot += m;
  }
  const float ms = tot / reps;
  printf("tile kinds: %d threads x %d, %d WG/CU, LDS table %d slots (%zu KB), %u %% LDS tiles: %.3f ms %.1f Ggather/s\n", TPBX, NPTX, wg_per_cu,
         LDSK, dynb / 1024, pct, ms, n / ms / 1e6);
  CK(hipFree(idx)); CK(hipFree(vals)); CK(hipFree(table)); CK(hipFree(out)); CK(hipFree(heads));
  return 0;
}

__global__ void fill_idx(uint32_t* idx, uint64_t n, uint32_t tiles_per_slab, uint64_t slice, uint64_t seed) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t slab = i / ((uint64_t)tiles_per_slab * TILE);
  idx[i] = (uint32_t)(slab * slice + mix(i + seed) % slice);
}

template <class T>
int run(double slice_mb, int wg_per_cu) {
  const uint32_t tiles_per_slab = 4096;
  const uint64_t n = (uint64_t)tiles_per_slab * 8 * TILE;  // 134M gathers
  const uint64_t slice = (uint64_t)(slice_mb * 1048576 / sizeof(T));
  uint32_t *idx, *heads, *census; T *table, *out;
  const unsigned grid = 256 * wg_per_cu;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&table, slice * 8 * sizeof(T))); CK(hipMalloc(&out, (uint64_t)grid * TPB * sizeof(T)));
  CK(hipMalloc(&heads, 64)); CK(hipMalloc(&census, 256));
  CK(hipMemset(table, 0, slice * 8 * sizeof(T))); CK(hipMemset(census, 0, 256));
  fill_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, tiles_per_slab, slice, 999);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms[2];
  for (int mode = 0; mode < 2; ++mode) {
    const int reps = 5;
    float tot = 0;
    for (int r = -2; r < reps; ++r) {
      CK(hipMemsetAsync(heads, 0, 64, 0));
      CK(hipEventRecord(a));
      if (mode == 0) gather_kernel<T, 0><<<grid, TPB>>>(idx, table, out, tiles_per_slab, heads, r == -2 ? census : nullptr);
      else gather_kernel<T, 1><<<grid, TPB>>>(idx, table, out, tiles_per_slab, heads, nullptr);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float m; CK(hipEventElapsedTime(&m, a, b));
      if (r >= 0) tot += m;
    }
    ms[mode] = tot / reps;
  }
  uint32_t h[64]; CK(hipMemcpy(h, census, 256, hipMemcpyDeviceToHost));
  uint32_t per_xcc[8] = {0}; int diag = 1;
  for (int x = 0; x < 8; ++x) { int nz = 0; for (int m = 0; m < 8; ++m) { per_xcc[x] += h[x * 8 + m]; nz += h[x * 8 + m] != 0; } if (nz != 1) diag = 0; }
  printf("elem %zuB slice %.2f MB x 8 = %6.1f MB, %d WG/CU: own-XCD slices %.3f ms %.1f Ggather/s | any XCD %.3f ms %.1f Ggather/s | WGs per XCD %u %u %u %u %u %u %u %u, blockIdx%%8 <-> XCD one-to-one: %s\n",
         sizeof(T), slice_mb, slice_mb * 8, wg_per_cu, ms[0], n / ms[0] / 1e6, ms[1], n / ms[1] / 1e6, per_xcc[0], per_xcc[1], per_xcc[2],
         per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7], diag ? "yes" : "no");
  CK(hipFree(idx)); CK(hipFree(table)); CK(hipFree(out)); CK(hipFree(heads)); CK(hipFree(census));
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 3) {  // fourth series: pure-LDS tiles next to table tiles
    for (uint32_t pct : {0u, 33u, 50u, 60u}) {
      run3<double, 256, 8, 2048>(2.75, 4, pct);
      run3<double, 512, 8, 4096>(2.75, 2, pct);
      run3<double, 512, 8, 8192>(2.75, 1, pct);
      run3<double, 1024, 8, 8192>(2.75, 1, pct);
      run3<double, 1024, 8, 12288>(2.75, 1, pct);
    }
    return 0;
  }
  if (argc > 2) {  // third series: a share of the gathers served from a 2048-slot LDS table (skew -P = P percent)
    for (int pct : {0, 30, 50, 70}) {
      run2<double, true, true, 0>(2.75, 3, -pct);
      run2<double, true, true, 2048>(2.75, 3, -pct);
    }
    return 0;
  }
  if (argc > 1) {  // second series: morph the loop towards the SpMV tile
    for (int skew : {0, 19}) {
      run2<double, false, false>(2.75, 4, skew);
      run2<double, true, false>(2.75, 4, skew);
      run2<double, true, true>(2.75, 4, skew);
      run2<double, true, true>(2.75, 3, skew);
    }
    run2<double, true, true>(1.0, 4, 19);
    run2<double, true, true>(2.0, 4, 19);
    run2<float, true, true>(1.0, 4, 0);
    return 0;
  }
  for (double mb : {1.0, 2.0, 3.0, 3.5, 4.0}) run<double>(mb, 8);
  run<double>(3.0, 4);
  run<double>(3.0, 2);
  for (double mb : {1.0, 3.0}) run<float>(mb, 8);
  return 0;
}
