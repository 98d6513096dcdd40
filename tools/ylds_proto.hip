// Prototype (diagnostic, not product): SpMV with the y accumulators of a row block in LDS and the block's
// entries streamed in COLUMN order, so that the x gathers of consecutive wave instructions fall into
// the same / neighbouring 128-byte lines (served by the CU's L1 instead of one L2 request per lane).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -munsafe-fp-atomics -o tools/libylds_proto.so tools/ylds_proto.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef const double __attribute__((address_space(1))) * gdp;
typedef const uint32_t __attribute__((address_space(1))) * gup;

struct Plan {
  const uint64_t* blk_ent;   // [nb+1] entry offsets (multiples of 256)
  const uint32_t* blk_acc;   // [nb+1] accumulator offsets
  const uint32_t* ecol;      // padded entries: column (or x index)
  const uint16_t* eacc;      // block-local accumulator
  const double* eval;        // value
  const uint32_t* acc_row;   // per accumulator: row id of a head accumulator, 0xFFFFFFFF otherwise
  uint32_t nb;
  unsigned long long* times;   // [nb][4] wall clock (100 MHz): start, zeroed, streamed, written; may be null
};

template <typename T> __device__ __forceinline__ T ntload(const T* p) { return __builtin_nontemporal_load(p); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct Stage {
  u32x4 c;
  u32x2 a;   // 4 x u16
  double v0, v1, v2, v3;
};

__device__ __forceinline__ void stage_load(Stage& s, const Plan& P, uint64_t base) {
  const u32x4* pc = reinterpret_cast<const u32x4*>(P.ecol + base);
  const u32x2* pa = reinterpret_cast<const u32x2*>(P.eacc + base);
  const f64x2* pv = reinterpret_cast<const f64x2*>(P.eval + base);
  s.c = ntload(pc);
  s.a = ntload(pa);
  f64x2 a = ntload(pv), b = ntload(pv + 1);
  s.v0 = a.x; s.v1 = a.y; s.v2 = b.x; s.v3 = b.y;
}

template <int NT, int ABL>
__global__ __launch_bounds__(NT) void ylds_kernel(Plan P, const double* __restrict__ x, double* __restrict__ y,
                                                  uint32_t* ticket, int nontemporal_x) {
  extern __shared__ __align__(16) double acc[];
  __shared__ uint32_t s_blk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = NT / 64;
  for (;;) {
    if (threadIdx.x == 0) s_blk = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t b = s_blk;
    if (b >= P.nb) break;
    unsigned long long t0 = wall_clock64();
    const uint32_t a0 = P.blk_acc[b], na = P.blk_acc[b + 1] - a0;
    for (uint32_t i = threadIdx.x; i < na; i += NT) acc[i] = 0.0;
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    const uint64_t c0 = P.blk_ent[b] >> 8, c1 = P.blk_ent[b + 1] >> 8;
    // three-stage pipeline per wavefront: stream loads of chunk i+2, x gathers of chunk i+1, LDS adds of chunk i;
    // unrolled three times so that every stage keeps its registers (a rotation by moves waits for the loads)
    uint64_t c = c0 + wave;
    Stage s0, s1, s2;
    double x0[4], x1[4], x2[4];
    double racc = 0;
    const gdp xg = (gdp)x;
#define LOADS(S, CC) stage_load(S, P, ((CC) < c1 ? (CC) : c1 - 1) * 256 + lane * 4);
// the empty asm statements pin a value's definition where the stage uses it: pure arithmetic otherwise moves
// above the scheduling barriers at IR level and drags the wait for its loads with it
#define GATHER(X, S, CC) {                                                            \
    asm volatile("" : "+v"(S.c));                                                      \
    if (ABL & 1) { X[0] = S.c.x; X[1] = S.c.y; X[2] = S.c.z; X[3] = S.c.w; }            \
    else { X[0] = xg[S.c.x]; X[1] = xg[S.c.y]; X[2] = xg[S.c.z]; X[3] = xg[S.c.w]; } }
#define CONSUME(S, X)                                                                  \
  asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]));                   \
  if (ABL & 2) { racc += S.v0 * X[0] + S.v1 * X[1] + S.v2 * X[2] + S.v3 * X[3] + (double)(S.a.x ^ S.a.y); } \
  else {                                                                               \
  atomicAdd(&acc[S.a.x & 0xFFFF], S.v0 * X[0]);   \
  atomicAdd(&acc[S.a.x >> 16], S.v1 * X[1]);      \
  atomicAdd(&acc[S.a.y & 0xFFFF], S.v2 * X[2]);   \
  atomicAdd(&acc[S.a.y >> 16], S.v3 * X[3]); }
#define STEP(SA, XA, SB, XB, SC)                   \
  if (c >= c1) break;                             \
  LOADS(SC, c + 2 * nw)                           \
  __builtin_amdgcn_sched_barrier(0);              \
  GATHER(XB, SB, c + nw)                          \
  __builtin_amdgcn_sched_barrier(0);              \
  CONSUME(SA, XA)                                 \
  __builtin_amdgcn_sched_barrier(0);              \
  c += nw;
    if (c0 < c1) {
    LOADS(s0, c)
    LOADS(s1, c + nw)
    GATHER(x0, s0, c)
    for (;;) {
      STEP(s0, x0, s1, x1, s2)
      STEP(s1, x1, s2, x2, s0)
      STEP(s2, x2, s0, x0, s1)
    }
    }
#undef STEP
    if ((ABL & 2) && racc == 1.2345) acc[0] = racc;
    __syncthreads();
    unsigned long long t2 = wall_clock64();
    for (uint32_t i = threadIdx.x; i < na; i += NT) {
      uint32_t r = P.acc_row[a0 + i];
      if (r != 0xFFFFFFFFu) {
        double s = acc[i];
        for (uint32_t j = i + 1; j < na && P.acc_row[a0 + j] == 0xFFFFFFFFu; ++j) s += acc[j];
        y[r] = s;
      }
    }
    __syncthreads();
    if (P.times && threadIdx.x == 0) {
      unsigned long long* t = P.times + 4ull * b;
      t[0] = t0; t[1] = t1; t[2] = t2; t[3] = wall_clock64();
    }
  }
}

// Second form: no cross-iteration pipeline.  A wavefront issues the stream loads of K chunks back to back, then the
// 4K gathers, then the 4K LDS adds: K * 3.5 KB of stream per wavefront in flight, nothing for the compiler to reorder.
template <int NT, int K, int ABL>
__global__ __launch_bounds__(NT) void ylds2_kernel(Plan P, const double* __restrict__ x, double* __restrict__ y,
                                                   uint32_t* ticket) {
  extern __shared__ __align__(16) double acc[];
  __shared__ uint32_t s_blk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = NT / 64;
  const gdp xg = (gdp)x;
  for (;;) {
    if (threadIdx.x == 0) s_blk = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t b = s_blk;
    if (b >= P.nb) break;
    unsigned long long t0 = wall_clock64();
    const uint32_t a0 = P.blk_acc[b], na = P.blk_acc[b + 1] - a0;
    for (uint32_t i = threadIdx.x; i < na; i += NT) acc[i] = 0.0;
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    const uint64_t c0 = P.blk_ent[b] >> 8, c1 = P.blk_ent[b + 1] >> 8;
    double racc = 0;
    uint64_t c = c0 + wave;
    for (; c + (uint64_t)(K - 1) * nw < c1; c += (uint64_t)K * nw) {
      Stage s[K];
      double xv[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k) stage_load(s[k], P, (c + (uint64_t)k * nw) * 256 + lane * 4);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (ABL & 1) { xv[k][0] = s[k].c.x; xv[k][1] = s[k].c.y; xv[k][2] = s[k].c.z; xv[k][3] = s[k].c.w; }
        else { xv[k][0] = xg[s[k].c.x]; xv[k][1] = xg[s[k].c.y]; xv[k][2] = xg[s[k].c.z]; xv[k][3] = xg[s[k].c.w]; }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (ABL & 2) {
          racc += s[k].v0 * xv[k][0] + s[k].v1 * xv[k][1] + s[k].v2 * xv[k][2] + s[k].v3 * xv[k][3] + (double)(s[k].a.x ^ s[k].a.y);
        } else {
          atomicAdd(&acc[s[k].a.x & 0xFFFF], s[k].v0 * xv[k][0]);
          atomicAdd(&acc[s[k].a.x >> 16], s[k].v1 * xv[k][1]);
          atomicAdd(&acc[s[k].a.y & 0xFFFF], s[k].v2 * xv[k][2]);
          atomicAdd(&acc[s[k].a.y >> 16], s[k].v3 * xv[k][3]);
        }
      }
    }
    for (; c < c1; c += nw) {
      Stage s1;
      stage_load(s1, P, c * 256 + lane * 4);
      double q0 = xg[s1.c.x], q1 = xg[s1.c.y], q2 = xg[s1.c.z], q3 = xg[s1.c.w];
      atomicAdd(&acc[s1.a.x & 0xFFFF], s1.v0 * q0);
      atomicAdd(&acc[s1.a.x >> 16], s1.v1 * q1);
      atomicAdd(&acc[s1.a.y & 0xFFFF], s1.v2 * q2);
      atomicAdd(&acc[s1.a.y >> 16], s1.v3 * q3);
    }
    if ((ABL & 2) && racc == 1.2345) acc[0] = racc;
    __syncthreads();
    unsigned long long t2 = wall_clock64();
    for (uint32_t i = threadIdx.x; i < na; i += NT) {
      uint32_t r = P.acc_row[a0 + i];
      if (r != 0xFFFFFFFFu) {
        double sum = acc[i];
        for (uint32_t j = i + 1; j < na && P.acc_row[a0 + j] == 0xFFFFFFFFu; ++j) sum += acc[j];
        y[r] = sum;
      }
    }
    __syncthreads();
    if (P.times && threadIdx.x == 0) {
      unsigned long long* t = P.times + 4ull * b;
      t[0] = t0; t[1] = t1; t[2] = t2; t[3] = wall_clock64();
    }
  }
}

// Third form: XCD-sliced.  Part p < 8 of the long rows gathers from slab table p (resident in the L2 of XCD p), part 8
// (cold columns) from x itself.  A block = (part, row group): the group's accumulators in LDS, the block's entries in slot
// order; it writes the partial sums partial[part][rank] of every row of the group.  Blocks of part p wait in queue p; a
// workgroup serves the queue of the XCD it runs on, then the cold queue, then whatever other queues still hold.
struct Plan3 {
  const uint64_t* blk_ent;   // [nb+1], blocks ordered part-major
  const uint32_t* blk_grp;   // [nb] row group of the block
  const uint32_t* grp_acc;   // [ng+1] accumulator offsets of the groups
  const uint32_t* acc_rank;  // per accumulator: long-row rank of a head accumulator, 0xFFFFFFFF otherwise
  const uint32_t* ecol;
  const uint16_t* eacc;
  const double* eval;
  const double* base[9];     // gather source of every part
  uint32_t part_blk[10];     // first block of every part
  uint64_t nl;               // long rows (stride of partial)
  double* partial;           // [9][nl]
  uint32_t* queue;           // [9] tickets, 64 B apart (16 u32)
  unsigned long long* times;
};

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7;
}

template <int NT, int K, int ABL>
__global__ __launch_bounds__(NT) void ylds3_kernel(Plan3 P) {
  extern __shared__ __align__(16) double acc[];
  __shared__ uint32_t s_blk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = NT / 64;
  const uint32_t xcc = xcc_id();
  int q = 0;  // position in this workgroup's queue order: own XCD, cold, then the others
  for (;;) {
    uint32_t part = q == 0 ? xcc : q == 1 ? 8u : ((xcc + (uint32_t)(q - 1)) & 7u);
    if (threadIdx.x == 0) s_blk = atomicAdd(P.queue + 16 * part, 1u);
    __syncthreads();
    uint32_t b = P.part_blk[part] + s_blk;
    __syncthreads();
    if (b >= P.part_blk[part + 1]) {
      if (++q > 8) break;
      continue;
    }
    unsigned long long t0 = wall_clock64();
    const uint32_t g = P.blk_grp[b];
    const uint32_t a0 = P.grp_acc[g], na = P.grp_acc[g + 1] - a0;
    for (uint32_t i = threadIdx.x; i < na; i += NT) acc[i] = 0.0;
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    const gdp xg = (gdp)P.base[part];
    const uint64_t c0 = P.blk_ent[b] >> 8, c1 = P.blk_ent[b + 1] >> 8;
    double racc = 0;
    uint64_t c = c0 + wave;
    Plan PP{nullptr, nullptr, P.ecol, P.eacc, P.eval, nullptr, 0, nullptr};
    for (; c + (uint64_t)(K - 1) * nw < c1; c += (uint64_t)K * nw) {
      Stage s[K];
      double xv[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k) stage_load(s[k], PP, (c + (uint64_t)k * nw) * 256 + lane * 4);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (ABL & 1) { xv[k][0] = s[k].c.x; xv[k][1] = s[k].c.y; xv[k][2] = s[k].c.z; xv[k][3] = s[k].c.w; }
        else { xv[k][0] = xg[s[k].c.x]; xv[k][1] = xg[s[k].c.y]; xv[k][2] = xg[s[k].c.z]; xv[k][3] = xg[s[k].c.w]; }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (ABL & 2) {
          racc += s[k].v0 * xv[k][0] + s[k].v1 * xv[k][1] + s[k].v2 * xv[k][2] + s[k].v3 * xv[k][3] + (double)(s[k].a.x ^ s[k].a.y);
        } else {
          atomicAdd(&acc[s[k].a.x & 0xFFFF], s[k].v0 * xv[k][0]);
          atomicAdd(&acc[s[k].a.x >> 16], s[k].v1 * xv[k][1]);
          atomicAdd(&acc[s[k].a.y & 0xFFFF], s[k].v2 * xv[k][2]);
          atomicAdd(&acc[s[k].a.y >> 16], s[k].v3 * xv[k][3]);
        }
      }
    }
    for (; c < c1; c += nw) {
      Stage s1;
      stage_load(s1, PP, c * 256 + lane * 4);
      double q0 = xg[s1.c.x], q1 = xg[s1.c.y], q2 = xg[s1.c.z], q3 = xg[s1.c.w];
      atomicAdd(&acc[s1.a.x & 0xFFFF], s1.v0 * q0);
      atomicAdd(&acc[s1.a.x >> 16], s1.v1 * q1);
      atomicAdd(&acc[s1.a.y & 0xFFFF], s1.v2 * q2);
      atomicAdd(&acc[s1.a.y >> 16], s1.v3 * q3);
    }
    if ((ABL & 2) && racc == 1.2345) acc[0] = racc;
    __syncthreads();
    unsigned long long t2 = wall_clock64();
    double* out = P.partial + (uint64_t)part * P.nl;
    for (uint32_t i = threadIdx.x; i < na; i += NT) {
      uint32_t r = P.acc_rank[a0 + i];
      if (r != 0xFFFFFFFFu) {
        double sum = acc[i];
        for (uint32_t j = i + 1; j < na && P.acc_rank[a0 + j] == 0xFFFFFFFFu; ++j) sum += acc[j];
        out[r] = sum;
      }
    }
    __syncthreads();
    if (P.times && threadIdx.x == 0) {
      unsigned long long* t = P.times + 4ull * b;
      t[0] = t0; t[1] = t1; t[2] = t2; t[3] = wall_clock64() | ((unsigned long long)xcc << 60);
    }
  }
}

extern "C" int ylds3_run(const uint64_t* blk_ent, const uint32_t* blk_grp, const uint32_t* grp_acc, const uint32_t* acc_rank,
                         const uint32_t* ecol, const uint16_t* eacc, const double* eval, const double* xh, uint64_t S,
                         const double* x, const uint32_t* part_blk, uint64_t nl, double* partial, uint32_t* queue,
                         unsigned long long* times, int nt, int kk, int abl, int lds_bytes, int grid, int iters, float* ms_out) {
  Plan3 P;
  P.blk_ent = blk_ent; P.blk_grp = blk_grp; P.grp_acc = grp_acc; P.acc_rank = acc_rank;
  P.ecol = ecol; P.eacc = eacc; P.eval = eval;
  for (int p = 0; p < 8; ++p) P.base[p] = xh + (uint64_t)p * S;
  P.base[8] = x;
  for (int p = 0; p < 10; ++p) P.part_blk[p] = part_blk[p];
  P.nl = nl; P.partial = partial; P.queue = queue; P.times = times;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() -> hipError_t {
    hipMemsetAsync(queue, 0, 9 * 64, 0);
#define LAUNCH3(NT_, K_, ABL_)                                                                                          \
  {                                                                                                                     \
    hipFuncSetAttribute((const void*)ylds3_kernel<NT_, K_, ABL_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
    ylds3_kernel<NT_, K_, ABL_><<<grid, NT_, lds_bytes, 0>>>(P);                                                         \
  }
    if (nt == 1024 && kk == 4 && abl == 0) LAUNCH3(1024, 4, 0)
    else if (nt == 1024 && kk == 4 && abl == 1) LAUNCH3(1024, 4, 1)
    else if (nt == 1024 && kk == 4 && abl == 2) LAUNCH3(1024, 4, 2)
    else if (nt == 1024 && kk == 2 && abl == 0) LAUNCH3(1024, 2, 0)
    else if (nt == 512 && kk == 4 && abl == 0) LAUNCH3(512, 4, 0)
    else if (nt == 512 && kk == 2 && abl == 0) LAUNCH3(512, 2, 0)
    else if (nt == 256 && kk == 4 && abl == 0) LAUNCH3(256, 4, 0)
    else return hipErrorInvalidValue;
    return hipGetLastError();
  };
  hipError_t err = launch();
  if (err != hipSuccess) return (int)err;
  err = hipDeviceSynchronize();
  if (err != hipSuccess) return (int)err;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0);
  err = hipEventSynchronize(e1);
  if (err != hipSuccess) return (int)err;
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / (iters > 0 ? iters : 1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

// Fourth form: as the third, with the entry stream of a block stored chunk by chunk (256 entries):
//   narrow parts (tables): u32 word[256] = slot << abits | acc, then T val[256]                      = 3072 B per chunk
//   wide part (cold):      u32 col[256], u16 acc[256], T val[256]                                    = 3584 B per chunk
// inside a chunk sorted entry w sits with lane w % 64 as its k = w / 64 th entry: word / col / acc at 4 * lane + k,
// val at (k / 2) * 128 + 2 * lane + k % 2 (so that every load instruction of a wavefront reads 1 KB contiguous).
struct Plan4 {
  const uint64_t* blk_off;   // [nb+1] byte offset of the block's first chunk in `stream` (blocks part-major)
  const uint32_t* blk_grp;
  const uint32_t* grp_acc;
  const uint32_t* acc_rank;
  const unsigned char* stream;
  const double* base[9];
  uint32_t part_blk[10];
  uint64_t nl;
  double* partial;
  uint32_t* queue;
  unsigned long long* times;
  uint32_t abits;
  uint32_t cold_first_mask;  // workgroups with (blockIdx.x >> 3) & mask == 0 start with the cold queue
};

template <int NT, int K, bool WIDE, bool NTL>
__device__ __forceinline__ void block_stream(const unsigned char* p0, uint64_t nchunks, int wave, int lane, gdp xg,
                                             double* acc, uint32_t abits) {
  constexpr int nw = NT / 64;
  constexpr uint64_t CB = WIDE ? 3584 : 3072;
  const uint32_t amask = (1u << abits) - 1;
  auto ld4 = [](const void* p) { return NTL ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p; };
  auto ld2 = [](const void* p) { return NTL ? __builtin_nontemporal_load((const u32x2*)p) : *(const u32x2*)p; };
  auto ldv = [](const void* p) { return NTL ? __builtin_nontemporal_load((const f64x2*)p) : *(const f64x2*)p; };
  uint64_t c = wave;
  for (; c + (uint64_t)(K - 1) * nw < nchunks; c += (uint64_t)K * nw) {
    u32x4 w[K];
    u32x2 a[K];
    f64x2 v0[K], v1[K];
    double xv[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned char* q = p0 + (c + (uint64_t)k * nw) * CB;
      w[k] = ld4(q + 16 * lane);
      if (WIDE) a[k] = ld2(q + 1024 + 8 * lane);
      const unsigned char* qv = q + (WIDE ? 1536 : 1024);
      v0[k] = ldv(qv + 16 * lane);
      v1[k] = ldv(qv + 1024 + 16 * lane);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (WIDE) { xv[k][0] = xg[w[k].x]; xv[k][1] = xg[w[k].y]; xv[k][2] = xg[w[k].z]; xv[k][3] = xg[w[k].w]; }
      else { xv[k][0] = xg[w[k].x >> abits]; xv[k][1] = xg[w[k].y >> abits]; xv[k][2] = xg[w[k].z >> abits]; xv[k][3] = xg[w[k].w >> abits]; }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      uint32_t a0, a1, a2, a3;
      if (WIDE) { a0 = a[k].x & 0xFFFF; a1 = a[k].x >> 16; a2 = a[k].y & 0xFFFF; a3 = a[k].y >> 16; }
      else { a0 = w[k].x & amask; a1 = w[k].y & amask; a2 = w[k].z & amask; a3 = w[k].w & amask; }
      atomicAdd(&acc[a0], v0[k].x * xv[k][0]);
      atomicAdd(&acc[a1], v0[k].y * xv[k][1]);
      atomicAdd(&acc[a2], v1[k].x * xv[k][2]);
      atomicAdd(&acc[a3], v1[k].y * xv[k][3]);
    }
  }
  for (; c < nchunks; c += nw) {
    const unsigned char* q = p0 + c * CB;
    u32x4 w = ld4(q + 16 * lane);
    u32x2 a = {0, 0};
    if (WIDE) a = ld2(q + 1024 + 8 * lane);
    const unsigned char* qv = q + (WIDE ? 1536 : 1024);
    f64x2 v0 = ldv(qv + 16 * lane), v1 = ldv(qv + 1024 + 16 * lane);
    double x0, x1, x2, x3;
    uint32_t a0, a1, a2, a3;
    if (WIDE) {
      x0 = xg[w.x]; x1 = xg[w.y]; x2 = xg[w.z]; x3 = xg[w.w];
      a0 = a.x & 0xFFFF; a1 = a.x >> 16; a2 = a.y & 0xFFFF; a3 = a.y >> 16;
    } else {
      x0 = xg[w.x >> abits]; x1 = xg[w.y >> abits]; x2 = xg[w.z >> abits]; x3 = xg[w.w >> abits];
      a0 = w.x & amask; a1 = w.y & amask; a2 = w.z & amask; a3 = w.w & amask;
    }
    atomicAdd(&acc[a0], v0.x * x0);
    atomicAdd(&acc[a1], v0.y * x1);
    atomicAdd(&acc[a2], v1.x * x2);
    atomicAdd(&acc[a3], v1.y * x3);
  }
}

template <int NT, int K, bool NTL>
__global__ __launch_bounds__(NT) void ylds4_kernel(Plan4 P) {
  extern __shared__ __align__(16) double acc[];
  __shared__ uint32_t s_blk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t xcc = xcc_id();
  const bool cold_first = ((blockIdx.x >> 3) & P.cold_first_mask) == 0;
  int q = 0;
  for (;;) {
    uint32_t part;
    if (cold_first) part = q == 0 ? 8u : q == 1 ? xcc : ((xcc + (uint32_t)(q - 1)) & 7u);
    else part = q == 0 ? xcc : q == 1 ? 8u : ((xcc + (uint32_t)(q - 1)) & 7u);
    if (threadIdx.x == 0) s_blk = atomicAdd(P.queue + 16 * part, 1u);
    __syncthreads();
    uint32_t b = P.part_blk[part] + s_blk;
    __syncthreads();
    if (b >= P.part_blk[part + 1]) {
      if (++q > 8) break;
      continue;
    }
    unsigned long long t0 = wall_clock64();
    const uint32_t g = P.blk_grp[b];
    const uint32_t a0 = P.grp_acc[g], na = P.grp_acc[g + 1] - a0;
    for (uint32_t i = threadIdx.x; i < na; i += NT) acc[i] = 0.0;
    __syncthreads();
    unsigned long long t1 = wall_clock64();
    const uint64_t o0 = P.blk_off[b], o1 = P.blk_off[b + 1];
    if (part == 8) block_stream<NT, K, true, NTL>(P.stream + o0, (o1 - o0) / 3584, wave, lane, (gdp)P.base[8], acc, P.abits);
    else block_stream<NT, K, false, NTL>(P.stream + o0, (o1 - o0) / 3072, wave, lane, (gdp)P.base[part], acc, P.abits);
    __syncthreads();
    unsigned long long t2 = wall_clock64();
    double* out = P.partial + (uint64_t)part * P.nl;
    for (uint32_t i = threadIdx.x; i < na; i += NT) {
      uint32_t r = P.acc_rank[a0 + i];
      if (r != 0xFFFFFFFFu) {
        double sum = acc[i];
        for (uint32_t j = i + 1; j < na && P.acc_rank[a0 + j] == 0xFFFFFFFFu; ++j) sum += acc[j];
        out[r] = sum;
      }
    }
    __syncthreads();
    if (P.times && threadIdx.x == 0) {
      unsigned long long* t = P.times + 4ull * b;
      t[0] = t0; t[1] = t1; t[2] = t2; t[3] = wall_clock64() | ((unsigned long long)xcc << 60);
    }
  }
}

extern "C" int ylds4_run(const uint64_t* blk_off, const uint32_t* blk_grp, const uint32_t* grp_acc, const uint32_t* acc_rank,
                         const unsigned char* stream, const double* xh, uint64_t S, const double* x, const uint32_t* part_blk,
                         uint64_t nl, double* partial, uint32_t* queue, unsigned long long* times, uint32_t abits,
                         uint32_t cold_first_mask, int nt, int kk, int ntl, int lds_bytes, int grid, int iters, float* ms_out) {
  Plan4 P;
  P.blk_off = blk_off; P.blk_grp = blk_grp; P.grp_acc = grp_acc; P.acc_rank = acc_rank; P.stream = stream;
  for (int p = 0; p < 8; ++p) P.base[p] = xh + (uint64_t)p * S;
  P.base[8] = x;
  for (int p = 0; p < 10; ++p) P.part_blk[p] = part_blk[p];
  P.nl = nl; P.partial = partial; P.queue = queue; P.times = times; P.abits = abits; P.cold_first_mask = cold_first_mask;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() -> hipError_t {
    hipMemsetAsync(queue, 0, 9 * 64, 0);
#define LAUNCH4(NT_, K_, NTL_)                                                                                          \
  {                                                                                                                     \
    hipFuncSetAttribute((const void*)ylds4_kernel<NT_, K_, NTL_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
    ylds4_kernel<NT_, K_, NTL_><<<grid, NT_, lds_bytes, 0>>>(P);                                                         \
  }
    if (nt == 1024 && kk == 4 && ntl) LAUNCH4(1024, 4, true)
    else if (nt == 1024 && kk == 2 && ntl) LAUNCH4(1024, 2, true)
    else if (nt == 1024 && kk == 2 && !ntl) LAUNCH4(1024, 2, false)
    else if (nt == 1024 && kk == 3 && ntl) LAUNCH4(1024, 3, true)
    else if (nt == 512 && kk == 4 && ntl) LAUNCH4(512, 4, true)
    else if (nt == 512 && kk == 2 && ntl) LAUNCH4(512, 2, true)
    else return hipErrorInvalidValue;
    return hipGetLastError();
  };
  hipError_t err = launch();
  if (err != hipSuccess) return (int)err;
  err = hipDeviceSynchronize();
  if (err != hipSuccess) return (int)err;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0);
  err = hipEventSynchronize(e1);
  if (err != hipSuccess) return (int)err;
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / (iters > 0 ? iters : 1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}

extern "C" int ylds_run(const uint64_t* blk_ent, const uint32_t* blk_acc, const uint32_t* ecol, const uint16_t* eacc,
                        const double* eval, const uint32_t* acc_row, uint32_t nb, const double* x, double* y,
                        uint64_t rows, uint32_t* ticket, int nt, int lds_bytes, int grid, int iters, float* ms_out,
                        unsigned long long* times, int zero_y, int abl, int kk) {
  Plan P{blk_ent, blk_acc, ecol, eacc, eval, acc_row, nb, times};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() -> hipError_t {
    hipMemsetAsync(ticket, 0, 4, 0);
    if (zero_y) hipMemsetAsync(y, 0, rows * 8, 0);
#define LAUNCH(NT_, ABL_)                                                                                           \
  {                                                                                                                 \
    hipFuncSetAttribute((const void*)ylds_kernel<NT_, ABL_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
    ylds_kernel<NT_, ABL_><<<grid, NT_, lds_bytes, 0>>>(P, x, y, ticket, 0);                                         \
  }
#define LAUNCH2(NT_, K_, ABL_)                                                                                          \
  {                                                                                                                     \
    hipFuncSetAttribute((const void*)ylds2_kernel<NT_, K_, ABL_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
    ylds2_kernel<NT_, K_, ABL_><<<grid, NT_, lds_bytes, 0>>>(P, x, y, ticket);                                           \
  }
    if (kk == 4 && nt == 1024 && abl == 0) LAUNCH2(1024, 4, 0)
    else if (kk == 4 && nt == 1024 && abl == 1) LAUNCH2(1024, 4, 1)
    else if (kk == 4 && nt == 1024 && abl == 2) LAUNCH2(1024, 4, 2)
    else if (kk == 4 && nt == 1024 && abl == 3) LAUNCH2(1024, 4, 3)
    else if (kk == 2 && nt == 1024) LAUNCH2(1024, 2, 0)
    else if (kk == 3 && nt == 1024) LAUNCH2(1024, 3, 0)
    else if (kk == 6 && nt == 1024) LAUNCH2(1024, 6, 0)
    else if (kk == 4 && nt == 512) LAUNCH2(512, 4, 0)
    else if (kk == 8 && nt == 512) LAUNCH2(512, 8, 0)
    else if (kk == 4 && nt == 256) LAUNCH2(256, 4, 0)
    else if (kk == 8 && nt == 256) LAUNCH2(256, 8, 0)
    else if (kk) return hipErrorInvalidValue;
    else if (nt == 1024 && abl == 0) LAUNCH(1024, 0)
    else if (nt == 1024 && abl == 1) LAUNCH(1024, 1)
    else if (nt == 1024 && abl == 2) LAUNCH(1024, 2)
    else if (nt == 1024 && abl == 3) LAUNCH(1024, 3)
    else if (nt == 512) LAUNCH(512, 0)
    else LAUNCH(256, 0)
    return hipGetLastError();
  };
  hipError_t err = launch();
  if (err != hipSuccess) return (int)err;
  err = hipDeviceSynchronize();
  if (err != hipSuccess) return (int)err;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0);
  err = hipEventSynchronize(e1);
  if (err != hipSuccess) return (int)err;
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / (iters > 0 ? iters : 1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}
