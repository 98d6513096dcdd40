// SpMV, XCD-sliced path, "row-group" form of the long rows (gfx950 only).
//
// The XCD-sliced analysis (spam_spmv.hip) deals the most referenced columns over eight table slabs, one per XCD's L2,
// and cuts every long row into nine PARTS: its entries in slab 0..7 and the cold rest.  The first form of that path
// kept each part as a CSR over the long rows and summed rows tile by tile: the x gathers of a tile go to unrelated
// table lines, every lane costs the L2 one request, and 245 G requests/s bound the kernel (0.78 ms for 154 M entries).
//
// This form turns the part around.  The long rows are ranked by length and cut into ROW GROUPS of at most 2^abits - 1
// accumulators; a BLOCK is (part, row group):
//   * the y accumulators of the group's rows live in LDS (8 bytes each: f64 for f32 / f64 values, u64 for i64);
//   * the block's entries are stored in TABLE-SLOT order, 256-entry chunks, each entry one 32-bit word
//     (slot << abits | accumulator) and its value: a wave instruction gathers 64 consecutive slots — a handful of
//     128-byte lines that the CU's L1 serves — and adds its products with LDS atomics (ds_add_f64 / ds_add_u64);
//   * the group's partial sums go to partial[part][rank of the row], dense and coalesced (accumulator j of a group is
//     the row of rank first + j), and sliced_reduce_kernel adds the nine parts into y[row of that rank].
// Rows that would draw most lanes of a wave instruction to one LDS address get extra accumulators behind the group's
// heads (an entry's accumulator is picked by its position in the row mod the count); they are folded into their head
// before the partial is written.
// Measured on the bench matrix (10 M x 10 M R-MAT, 153.7 M entries in 1.48 M long rows): 0.42 ms against 0.78 ms;
// without the gathers the kernel takes the same time — what is left is the stream itself (12 bytes per entry).
//
// Sums of f32 / f64 values are accumulated by LDS atomics in an order that varies from run to run: results are
// reproducible to rounding, not bit for bit (i64 results are exact and bit-identical).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "spam_internal.hpp"

namespace spam {

namespace {

constexpr uint32_t RG_NONE = 0xFFFFFFFFu;
constexpr int RG_KEY_ACC_BITS = 14, RG_KEY_IDX_BITS = 32;
constexpr int RG_KEY_BLK_SHIFT = RG_KEY_ACC_BITS + RG_KEY_IDX_BITS;  // 46
constexpr uint32_t RG_MAX_SUB = 512;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define RG_GLB __attribute__((address_space(1)))

template <class T>
struct RgAcc {
  typedef double type;
  static __device__ __forceinline__ double prod(T v, T x) { return (double)v * (double)x; }
};
template <>
struct RgAcc<i64w> {
  typedef i64w type;
  static __device__ __forceinline__ i64w prod(i64w v, i64w x) { return v * x; }
};

__device__ __forceinline__ uint32_t rg_xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7u;
}

template <class T>
__host__ __device__ constexpr uint32_t rg_chunk_bytes(bool wide) {
  return 1024u + (wide ? 512u : 0u) + 256u * (uint32_t)sizeof(T);
}
// position of the k-th entry of lane `lane` inside a chunk's value array
template <class T>
__host__ __device__ __forceinline__ uint32_t rg_val_slot(uint32_t lane, uint32_t k) {
  return sizeof(T) == 8 ? (k >> 1) * 128u + 2u * lane + (k & 1u) : 4u * lane + k;
}

template <class T>
struct RgArgs {
  const uint64_t* blk_off;    // [NPART * ng + 1] byte offset of every block's first chunk (part-major)
  const uint32_t* grp_row;    // [ng + 1] first rank of every group (its head accumulators are its rows, in rank order)
  const uint32_t* grp_sub;    // [ng + 1] first extra accumulator of every group in sub_head
  const uint32_t* sub_head;   // per extra accumulator: the head (local) it is folded into
  const unsigned char* stream;
  const T* base[SPMV_NPART];  // gather source of every part: the slab tables, x for the cold part
  T* partial;                 // [NPART][nl]
  uint32_t* queue;            // NPART heads, SPMV_HEAD_STRIDE words apart (reset by sliced_build_tables_kernel)
  uint32_t ng, nl, abits;
};

template <class T>
__device__ __forceinline__ void rg_load_vals(const unsigned char* qv, int lane, T (&v)[4]) {
  if constexpr (sizeof(T) == 8) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    const V2 a = __builtin_nontemporal_load((const V2*)(qv + 16 * lane));
    const V2 b = __builtin_nontemporal_load((const V2*)(qv + 1024 + 16 * lane));
    v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
  } else {
    typedef T V4 __attribute__((ext_vector_type(4)));
    const V4 a = __builtin_nontemporal_load((const V4*)(qv + 16 * lane));
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w;
  }
}

// One block's chunks, dealt round-robin to the wavefronts.  A wavefront issues the stream loads of K chunks back to
// back, then their 4 K gathers, then the 4 K LDS adds: K * 3 KB per wavefront in flight and straight-line code between
// the waits (a cross-iteration software pipeline was measured no faster, and the compiler's register rotation made it
// wait for the loads it had just issued).
template <class T, int NT, int K, bool WIDE>
__device__ __forceinline__ void rg_stream(const unsigned char* p0, uint64_t nchunks, int wave, int lane,
                                          const T RG_GLB* xg, typename RgAcc<T>::type* acc, uint32_t abits) {
  typedef RgAcc<T> A;
  constexpr int nw = NT / 64;
  constexpr uint32_t CB = rg_chunk_bytes<T>(WIDE);
  constexpr uint32_t VOFF = WIDE ? 1536u : 1024u;
  const uint32_t amask = (1u << abits) - 1u;
  uint64_t c = (uint64_t)wave;
  for (; c + (uint64_t)(K - 1) * nw < nchunks; c += (uint64_t)K * nw) {
    u32x4 w[K];
    u32x2 a[K];
    T v[K][4], xv[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned char* q = p0 + (c + (uint64_t)k * nw) * CB;
      w[k] = __builtin_nontemporal_load((const u32x4*)(q + 16 * lane));
      if (WIDE) a[k] = __builtin_nontemporal_load((const u32x2*)(q + 1024 + 8 * lane));
      rg_load_vals<T>(q + VOFF, lane, v[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (WIDE) {
        xv[k][0] = xg[w[k].x], xv[k][1] = xg[w[k].y], xv[k][2] = xg[w[k].z], xv[k][3] = xg[w[k].w];
      } else {
        xv[k][0] = xg[w[k].x >> abits], xv[k][1] = xg[w[k].y >> abits];
        xv[k][2] = xg[w[k].z >> abits], xv[k][3] = xg[w[k].w >> abits];
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      uint32_t a0, a1, a2, a3;
      if (WIDE) {
        a0 = a[k].x & 0xFFFFu, a1 = a[k].x >> 16, a2 = a[k].y & 0xFFFFu, a3 = a[k].y >> 16;
      } else {
        a0 = w[k].x & amask, a1 = w[k].y & amask, a2 = w[k].z & amask, a3 = w[k].w & amask;
      }
      atomicAdd(&acc[a0], A::prod(v[k][0], xv[k][0]));
      atomicAdd(&acc[a1], A::prod(v[k][1], xv[k][1]));
      atomicAdd(&acc[a2], A::prod(v[k][2], xv[k][2]));
      atomicAdd(&acc[a3], A::prod(v[k][3], xv[k][3]));
    }
  }
  for (; c < nchunks; c += nw) {
    const unsigned char* q = p0 + c * CB;
    const u32x4 w = __builtin_nontemporal_load((const u32x4*)(q + 16 * lane));
    u32x2 a = {0u, 0u};
    if (WIDE) a = __builtin_nontemporal_load((const u32x2*)(q + 1024 + 8 * lane));
    T v[4];
    rg_load_vals<T>(q + VOFF, lane, v);
    T x0, x1, x2, x3;
    uint32_t a0, a1, a2, a3;
    if (WIDE) {
      x0 = xg[w.x], x1 = xg[w.y], x2 = xg[w.z], x3 = xg[w.w];
      a0 = a.x & 0xFFFFu, a1 = a.x >> 16, a2 = a.y & 0xFFFFu, a3 = a.y >> 16;
    } else {
      x0 = xg[w.x >> abits], x1 = xg[w.y >> abits], x2 = xg[w.z >> abits], x3 = xg[w.w >> abits];
      a0 = w.x & amask, a1 = w.y & amask, a2 = w.z & amask, a3 = w.w & amask;
    }
    atomicAdd(&acc[a0], A::prod(v[0], x0));
    atomicAdd(&acc[a1], A::prod(v[1], x1));
    atomicAdd(&acc[a2], A::prod(v[2], x2));
    atomicAdd(&acc[a3], A::prod(v[3], x3));
  }
}

// Persistent workgroups.  A workgroup reads the XCD it runs on and serves that slab's queue, then the cold queue, then
// whatever the other slabs have left (so nothing depends on the placement); one workgroup in four starts with the cold
// queue, whose blocks wait for fabric round trips and overlap with the streaming of the others.
template <class T, int NT, int K>
__global__ __launch_bounds__(NT) void spmv_rg_kernel(RgArgs<T> P) {
  typedef typename RgAcc<T>::type AccT;
  extern __shared__ __align__(16) unsigned char rg_lds[];
  AccT* acc = reinterpret_cast<AccT*>(rg_lds);
  __shared__ uint32_t s_blk;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t xcc = rg_xcc_id();
  const bool cold_first = ((blockIdx.x >> 3) & 3u) == 0u;
  int q = 0;
  for (;;) {
    uint32_t part;
    if (cold_first) part = q == 0 ? (uint32_t)SPMV_NSLAB : q == 1 ? xcc : ((xcc + (uint32_t)(q - 1)) & 7u);
    else part = q == 0 ? xcc : q == 1 ? (uint32_t)SPMV_NSLAB : ((xcc + (uint32_t)(q - 1)) & 7u);
    if (threadIdx.x == 0) s_blk = atomicAdd(P.queue + (uint64_t)SPMV_HEAD_STRIDE * part, 1u);
    __syncthreads();
    const uint32_t g = s_blk;
    __syncthreads();
    if (g >= P.ng) {
      if (++q > SPMV_NSLAB) break;
      continue;
    }
    const uint64_t b = (uint64_t)part * P.ng + g;
    const uint64_t o0 = P.blk_off[b], o1 = P.blk_off[b + 1];
    if (o1 == o0) continue;  // no entry of this group in this part: its partial sums stay zero
    const uint32_t r0 = P.grp_row[g], nrows = P.grp_row[g + 1] - r0;
    const uint32_t s0 = P.grp_sub[g], nsubs = P.grp_sub[g + 1] - s0;
    const uint32_t na = nrows + nsubs;
    for (uint32_t i = threadIdx.x; i <= na; i += NT) acc[i] = AccT(0);  // [na] takes the padding entries
    __syncthreads();
    if (part == (uint32_t)SPMV_NSLAB)
      rg_stream<T, NT, K, true>(P.stream + o0, (o1 - o0) / rg_chunk_bytes<T>(true), wave, lane, (const T RG_GLB*)P.base[part],
                                acc, P.abits);
    else
      rg_stream<T, NT, K, false>(P.stream + o0, (o1 - o0) / rg_chunk_bytes<T>(false), wave, lane,
                                 (const T RG_GLB*)P.base[part], acc, P.abits);
    __syncthreads();
    if (nsubs) {  // uniform
      for (uint32_t i = threadIdx.x; i < nsubs; i += NT) atomicAdd(&acc[P.sub_head[s0 + i]], acc[nrows + i]);
      __syncthreads();
    }
    T* out = P.partial + (uint64_t)part * P.nl + r0;
    for (uint32_t i = threadIdx.x; i < nrows; i += NT) out[i] = (T)acc[i];
    __syncthreads();
  }
}

// ---- plan construction ------------------------------------------------------------------------------------------
__global__ void rg_len_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ long_rows, uint32_t nl,
                              uint32_t* __restrict__ len, uint32_t* __restrict__ key, uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nl) return;
  const uint32_t r = long_rows[i];
  const uint32_t l = row_ptr[r + 1] - row_ptr[r];
  len[i] = l;
  key[i] = ~l;  // ascending sort of ~len = descending lengths
  idx[i] = i;
}

// One wavefront per long row: sort key (block | table slot or column | accumulator) and source position of every entry.
__global__ void rg_keys_kernel(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                               const uint32_t* __restrict__ code, const uint8_t* __restrict__ slab,
                               const uint32_t* __restrict__ long_rows, uint32_t nl, const uint64_t* __restrict__ lofs,
                               const uint32_t* __restrict__ row_grp, const uint32_t* __restrict__ row_acc,
                               const uint32_t* __restrict__ row_sub, uint32_t ng, uint64_t* __restrict__ keys,
                               uint32_t* __restrict__ src) {
  const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (w >= nl) return;
  const uint32_t r = long_rows[w];
  const uint32_t s = row_ptr[r], e = row_ptr[r + 1];
  const uint32_t g = row_grp[w], ra = row_acc[w];
  const uint32_t head = ra & 0xFFFFu, nsub = ra >> 16;  // accumulators of the row: its head and nsub - 1 extra ones
  const uint32_t sbase = nsub > 1 ? row_sub[w] : 0u;
  const uint64_t o = lofs[w];
  for (uint32_t p = s + lane; p < e; p += 64) {
    const uint32_t c = col[p];
    const uint64_t b = (uint64_t)slab[c] * ng + g;
    const uint32_t sub = nsub > 1 ? (p - s) % nsub : 0u;
    const uint32_t acc = sub ? sbase + sub - 1u : head;
    keys[o + (p - s)] = (b << RG_KEY_BLK_SHIFT) | ((uint64_t)code[c] << RG_KEY_ACC_BITS) | (uint64_t)acc;
    src[o + (p - s)] = p;
  }
}

__global__ void rg_rank_rows_kernel(const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ idx_of_rank, uint32_t nl,
                                    uint32_t* __restrict__ rank_rows) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nl) rank_rows[k] = long_rows[idx_of_rank[k]];
}
__global__ void rg_fill_u64_kernel(uint64_t* __restrict__ p, uint64_t n, uint64_t v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void rg_bounds_kernel(const uint64_t* __restrict__ keys, uint64_t n, uint64_t* __restrict__ start) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t b = keys[j] >> RG_KEY_BLK_SHIFT;
  if (j == 0 || (keys[j - 1] >> RG_KEY_BLK_SHIFT) != b) start[b] = j;
}

// block of a padded entry position: the last b with pe[b] <= q
__device__ __forceinline__ uint32_t rg_find_block(const uint64_t* __restrict__ pe, uint32_t nb, uint64_t q) {
  uint32_t lo = 0, hi = nb;  // pe[lo] <= q < pe[hi]
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (pe[mid] <= q) lo = mid;
    else hi = mid;
  }
  return lo;
}

// One thread per padded entry position (in sorted order inside its chunk): write the entry word, accumulator and
// value at the position the kernel's lanes read them from, and remember where the value came from.
template <class T>
__global__ void rg_fill_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ src_sorted,
                               const uint64_t* __restrict__ start, const uint64_t* __restrict__ pe,
                               const uint64_t* __restrict__ blk_off, const uint32_t* __restrict__ grp_row,
                               const uint32_t* __restrict__ grp_sub, uint32_t nb,
                               uint32_t ng, uint32_t abits, const T* __restrict__ vals, unsigned char* __restrict__ stream,
                               uint32_t* __restrict__ vsrc, uint64_t padded) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= padded) return;
  const uint32_t b = rg_find_block(pe, nb, q);
  const uint32_t part = b / ng, g = b - part * ng;
  const bool wide = part == (uint32_t)SPMV_NSLAB;
  const uint64_t jb = q - pe[b];
  const uint64_t chunk = jb >> 8;
  const uint32_t w = (uint32_t)(jb & 255u), lane = w & 63u, k = w >> 6;
  const uint64_t cnt = start[b + 1] - start[b];
  // padding: the group's spare accumulator, behind its heads and extra accumulators
  uint32_t idx = 0, a = (grp_row[g + 1] - grp_row[g]) + (grp_sub[g + 1] - grp_sub[g]), p = RG_NONE;
  T v = T(0);
  if (jb < cnt) {
    const uint64_t key = keys[start[b] + jb];
    idx = (uint32_t)(key >> RG_KEY_ACC_BITS);
    a = (uint32_t)(key & ((1u << RG_KEY_ACC_BITS) - 1u));
    p = src_sorted[start[b] + jb];
    v = vals[p];
  }
  unsigned char* c = stream + blk_off[b] + chunk * (wide ? rg_chunk_bytes<T>(true) : rg_chunk_bytes<T>(false));
  const uint32_t wpos = 4u * lane + k;
  if (wide) {
    reinterpret_cast<uint32_t*>(c)[wpos] = idx;
    reinterpret_cast<uint16_t*>(c + 1024)[wpos] = (uint16_t)a;
  } else {
    reinterpret_cast<uint32_t*>(c)[wpos] = (idx << abits) | a;
  }
  const uint32_t vs = rg_val_slot<T>(lane, k);
  reinterpret_cast<T*>(c + (wide ? 1536 : 1024))[vs] = v;
  vsrc[pe[b] + (chunk << 8) + vs] = p;
}

// spam_csr_values_changed: one thread per padded value position
template <class T>
__global__ void rg_refresh_kernel(const uint64_t* __restrict__ pe, const uint64_t* __restrict__ blk_off, uint32_t nb, uint32_t ng,
                                  const uint32_t* __restrict__ vsrc, const T* __restrict__ vals,
                                  unsigned char* __restrict__ stream, uint64_t padded) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= padded) return;
  const uint32_t p = vsrc[q];
  if (p == RG_NONE) return;
  const uint32_t b = rg_find_block(pe, nb, q);
  const bool wide = b / ng == (uint32_t)SPMV_NSLAB;
  const uint64_t jb = q - pe[b];
  unsigned char* c = stream + blk_off[b] + (jb >> 8) * (wide ? rg_chunk_bytes<T>(true) : rg_chunk_bytes<T>(false));
  reinterpret_cast<T*>(c + (wide ? 1536 : 1024))[jb & 255u] = vals[p];
}

static int bits_for(uint64_t v) {  // bits needed to represent values 0..v
  int b = 1;
  while (b < 64 && (v >> b)) ++b;
  return b;
}

template <class T, int NT, int K>
static int rg_launch_or_grid(const RgPlan* g, const RgArgs<T>* args, hipStream_t s, unsigned* grid_out) {
  auto kern = spmv_rg_kernel<T, NT, K>;
  if (grid_out) {
    if (g->lds_bytes > 48 * 1024)
      SPAM_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds_bytes));
    int per_cu = 0, dev = 0, cus = 0;
    SPAM_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, g->lds_bytes));
    SPAM_HIP_TRY(hipGetDevice(&dev));
    SPAM_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (per_cu < 1) return set_error(SPAM_E_HIP, "row-group SpMV kernel does not fit a CU (%u B of LDS)", g->lds_bytes);
    uint64_t n = (uint64_t)per_cu * (uint64_t)cus;
    if (n > g->nonempty_blocks) n = g->nonempty_blocks;
    *grid_out = (unsigned)std::max<uint64_t>(n, 1);
    return SPAM_OK;
  }
  kern<<<g->grid, NT, g->lds_bytes, s>>>(*args);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

template <class T>
static int rg_dispatch(const RgPlan* g, const RgArgs<T>* args, hipStream_t s, unsigned* grid_out) {
#define SPAM_RG_VARIANT(NT_, K_) \
  if (g->nt == NT_ && g->k == K_) return rg_launch_or_grid<T, NT_, K_>(g, args, s, grid_out)
  SPAM_RG_VARIANT(512, 4);
  SPAM_RG_VARIANT(1024, 2);
  SPAM_RG_VARIANT(1024, 4);
  SPAM_RG_VARIANT(256, 4);
#undef SPAM_RG_VARIANT
  return set_error(SPAM_E_INVALID, "no row-group SpMV kernel for %d threads x %d chunks", g->nt, g->k);
}

}  // namespace

void rg_plan_free(RgPlan* g) {
  if (!g) return;
  void* ptrs[] = {g->stream, g->blk_off, g->pe, g->grp_row, g->grp_sub, g->sub_head, g->rank_rows, g->vsrc};
  for (void* p : ptrs)
    if (p) dev_free(p);
  delete g;
}

template <class T>
int rg_build(const spam_csr* a, const uint8_t* slab, const uint32_t* code, const uint32_t* long_rows, uint32_t nl, uint32_t S,
             const RgKnobs& kn, hipStream_t s, RgPlan** out) {
  *out = nullptr;
  if (nl == 0 || !a->row_ptr32) return SPAM_OK;
  // one word holds slot << abits | accumulator
  const int sbits = bits_for(S ? S - 1 : 0);
  int abits = std::min(13, 32 - sbits);
  if (kn.acc_bits > 0) abits = std::min(abits, kn.acc_bits);
  if (abits < 9) return SPAM_OK;  // tables too wide for the packed word: the caller keeps the tile form
  const uint32_t rmax = (1u << abits) - 1u;  // usable accumulators of a group (+ 1 spare for the padding entries)
  const uint64_t eblk = (uint64_t)std::max(1024, kn.eblk);
  const uint64_t ecap = eblk * SPMV_NPART;  // entries of a group over its nine blocks

  // -- long rows by length ------------------------------------------------------------------------------------------
  DevBuf len, key_in, key_out, idx_in, idx_out, tmp;
  SPAM_TRY(len.alloc((uint64_t)nl * 4));
  SPAM_TRY(key_in.alloc((uint64_t)nl * 4));
  SPAM_TRY(key_out.alloc((uint64_t)nl * 4));
  SPAM_TRY(idx_in.alloc((uint64_t)nl * 4));
  SPAM_TRY(idx_out.alloc((uint64_t)nl * 4));
  const unsigned gl = (nl + 255) / 256;
  rg_len_kernel<<<gl, 256, 0, s>>>(a->row_ptr32, long_rows, nl, len.as<uint32_t>(), key_in.as<uint32_t>(), idx_in.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  size_t tb = 0;
  SPAM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key_in.as<uint32_t>(), key_out.as<uint32_t>(), idx_in.as<uint32_t>(),
                                                  idx_out.as<uint32_t>(), (int64_t)nl, 0, 32, s));
  SPAM_TRY(tmp.alloc(tb));
  SPAM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, key_in.as<uint32_t>(), key_out.as<uint32_t>(), idx_in.as<uint32_t>(),
                                                  idx_out.as<uint32_t>(), (int64_t)nl, 0, 32, s));
  std::vector<uint32_t> h_key(nl), h_idx(nl);
  SPAM_HIP_TRY(hipMemcpyAsync(h_key.data(), key_out.p, (uint64_t)nl * 4, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipMemcpyAsync(h_idx.data(), idx_out.p, (uint64_t)nl * 4, hipMemcpyDeviceToHost, s));
  DevBuf lofs, stmp;
  SPAM_TRY(lofs.alloc(((uint64_t)nl + 1) * 8));
  SPAM_TRY(stmp.alloc(scan_tmp_bytes(nl)));
  SPAM_TRY(scan_u32_to_u64(len.as<uint32_t>(), nl, lofs.as<uint64_t>(), stmp.p, s));
  uint64_t n_ent = 0;
  SPAM_HIP_TRY(hipMemcpyAsync(&n_ent, lofs.as<uint64_t>() + nl, 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  if (n_ent == 0 || n_ent >= 0xFFFFFFFFull) return SPAM_OK;

  // -- row groups (host): consecutive ranks, at most rmax accumulators and ecap entries ---------------------------------
  std::vector<uint32_t> h_grp(nl), h_racc(nl), h_rsub(nl, 0u), h_grp_row(1, 0u), h_grp_sub(1, 0u), h_sub_head;
  {
    uint32_t g = 0, accs = 0, nrows = 0, nsubs = 0, first = 0;
    uint64_t ents = 0;
    std::vector<uint32_t> pending;  // (rank, extra accumulators) of the current group's rows that have some
    auto close = [&](uint32_t end_rank) {
      // the extra accumulators follow the group's heads: only now is the number of heads known
      uint32_t sb = nrows;
      for (size_t t = 0; t < pending.size(); t += 2) {
        const uint32_t k = pending[t], extra = pending[t + 1];
        h_rsub[h_idx[k]] = sb;
        for (uint32_t j = 0; j < extra; ++j) h_sub_head.push_back(k - first);
        sb += extra;
      }
      pending.clear();
      h_grp_row.push_back(end_rank);
      h_grp_sub.push_back((uint32_t)h_sub_head.size());
    };
    for (uint32_t k = 0; k < nl; ++k) {
      const uint64_t L = (uint64_t)(~h_key[k]);
      // a row holding the fraction f of a block's entries draws 64 f lanes of every wave instruction: 8 x as many
      // accumulators keep two lanes off one LDS address most of the time
      uint64_t nsub = (L * RG_MAX_SUB + ecap - 1) / ecap;
      nsub = std::min<uint64_t>(std::max<uint64_t>(nsub, 1), std::min<uint64_t>(RG_MAX_SUB, std::max<uint64_t>(L, 1)));
      if (accs + nsub > rmax || (ents > 0 && ents + L > ecap)) {
        close(k);
        ++g, accs = 0, ents = 0, nrows = 0, nsubs = 0, first = k;
      }
      const uint32_t i = h_idx[k];
      h_grp[i] = g;
      h_racc[i] = nrows | ((uint32_t)nsub << 16);
      if (nsub > 1) pending.push_back(k), pending.push_back((uint32_t)nsub - 1);
      accs += (uint32_t)nsub, ents += L, ++nrows, nsubs += (uint32_t)nsub - 1;
    }
    close(nl);
  }
  const uint32_t ng = (uint32_t)h_grp_row.size() - 1;
  const uint64_t nb64 = (uint64_t)ng * SPMV_NPART;
  const int bbits = bits_for(nb64 - 1);
  if (RG_KEY_BLK_SHIFT + bbits > 64) return SPAM_OK;
  const uint32_t nb = (uint32_t)nb64;

  RgPlan* g = new (std::nothrow) RgPlan();
  if (!g) return set_error(SPAM_E_OOM, "host allocation failed");
  struct Guard {
    RgPlan* g;
    ~Guard() { if (g) rg_plan_free(g); }
  } guard{g};
  g->ng = ng, g->abits = (uint32_t)abits, g->entries = n_ent, g->naccs = (uint64_t)nl + h_sub_head.size();
  g->nt = kn.nt, g->k = kn.k;
  g->lds_bytes = (rmax + 1u) * 8u;

  DevBuf row_grp, row_acc, row_sub, grp_row, grp_sub, sub_head, rank_rows;
  SPAM_TRY(row_grp.alloc((uint64_t)nl * 4));
  SPAM_TRY(row_acc.alloc((uint64_t)nl * 4));
  SPAM_TRY(row_sub.alloc((uint64_t)nl * 4));
  SPAM_TRY(grp_row.alloc(((uint64_t)ng + 1) * 4));
  SPAM_TRY(grp_sub.alloc(((uint64_t)ng + 1) * 4));
  SPAM_TRY(sub_head.alloc(h_sub_head.size() * 4 + 4));
  SPAM_TRY(rank_rows.alloc((uint64_t)nl * 4));
  SPAM_HIP_TRY(hipMemcpyAsync(row_grp.p, h_grp.data(), (uint64_t)nl * 4, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(row_acc.p, h_racc.data(), (uint64_t)nl * 4, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(row_sub.p, h_rsub.data(), (uint64_t)nl * 4, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(grp_row.p, h_grp_row.data(), ((uint64_t)ng + 1) * 4, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(grp_sub.p, h_grp_sub.data(), ((uint64_t)ng + 1) * 4, hipMemcpyHostToDevice, s));
  if (!h_sub_head.empty())
    SPAM_HIP_TRY(hipMemcpyAsync(sub_head.p, h_sub_head.data(), h_sub_head.size() * 4, hipMemcpyHostToDevice, s));
  rg_rank_rows_kernel<<<gl, 256, 0, s>>>(long_rows, idx_out.as<uint32_t>(), nl, rank_rows.as<uint32_t>());
  SPAM_CHECK_LAUNCH();

  // -- entries: key, sort by (block, slot) -------------------------------------------------------------------------------
  DevBuf keys_in, keys_out, src_in, src_out;
  SPAM_TRY(keys_in.alloc(n_ent * 8));
  SPAM_TRY(keys_out.alloc(n_ent * 8));
  SPAM_TRY(src_in.alloc(n_ent * 4));
  SPAM_TRY(src_out.alloc(n_ent * 4));
  const unsigned gw = (unsigned)(((uint64_t)nl * 64 + 255) / 256);
  rg_keys_kernel<<<gw, 256, 0, s>>>(a->row_ptr32, a->col_idx, code, slab, long_rows, nl, lofs.as<uint64_t>(), row_grp.as<uint32_t>(),
                                    row_acc.as<uint32_t>(), row_sub.as<uint32_t>(), ng, keys_in.as<uint64_t>(), src_in.as<uint32_t>());
  SPAM_CHECK_LAUNCH();
  const int end_bit = RG_KEY_BLK_SHIFT + bbits;
  tb = 0;
  SPAM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), src_in.as<uint32_t>(),
                                                  src_out.as<uint32_t>(), (int64_t)n_ent, RG_KEY_ACC_BITS, end_bit, s));
  SPAM_TRY(tmp.alloc(tb));
  SPAM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), src_in.as<uint32_t>(),
                                                  src_out.as<uint32_t>(), (int64_t)n_ent, RG_KEY_ACC_BITS, end_bit, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // the host vectors above are read by the copies
  keys_in.reset(), src_in.reset(), tmp.reset();

  // -- block bounds, chunk offsets ---------------------------------------------------------------------------------------
  DevBuf start;
  SPAM_TRY(start.alloc(((uint64_t)nb + 1) * 8));
  rg_fill_u64_kernel<<<(nb + 1 + 255) / 256, 256, 0, s>>>(start.as<uint64_t>(), (uint64_t)nb + 1, ~0ull);
  SPAM_CHECK_LAUNCH();
  rg_bounds_kernel<<<(unsigned)((n_ent + 255) / 256), 256, 0, s>>>(keys_out.as<uint64_t>(), n_ent, start.as<uint64_t>());
  SPAM_CHECK_LAUNCH();
  std::vector<uint64_t> h_start((size_t)nb + 1), h_off((size_t)nb + 1), h_pe((size_t)nb + 1);
  SPAM_HIP_TRY(hipMemcpyAsync(h_start.data(), start.p, ((uint64_t)nb + 1) * 8, hipMemcpyDeviceToHost, s));
  SPAM_HIP_TRY(hipStreamSynchronize(s));
  h_start[nb] = n_ent;
  for (uint32_t b = nb; b-- > 0;)
    if (h_start[b] == ~0ull) h_start[b] = h_start[b + 1];
  h_off[0] = 0, h_pe[0] = 0;
  uint64_t nonempty = 0;
  for (uint32_t b = 0; b < nb; ++b) {
    const uint64_t cnt = h_start[b + 1] - h_start[b], nch = (cnt + 255) / 256;
    const bool wide = b / ng == (uint32_t)SPMV_NSLAB;
    h_off[b + 1] = h_off[b] + nch * rg_chunk_bytes<T>(wide);
    h_pe[b + 1] = h_pe[b] + nch * 256;
    nonempty += cnt != 0;
  }
  g->stream_bytes = h_off[nb], g->padded = h_pe[nb], g->nonempty_blocks = nonempty;
  DevBuf blk_off, pe, stream, vsrc;
  SPAM_TRY(blk_off.alloc(((uint64_t)nb + 1) * 8));
  SPAM_TRY(pe.alloc(((uint64_t)nb + 1) * 8));
  SPAM_TRY(stream.alloc(g->stream_bytes + 16));
  SPAM_TRY(vsrc.alloc(g->padded * 4 + 16));
  SPAM_HIP_TRY(hipMemcpyAsync(start.p, h_start.data(), ((uint64_t)nb + 1) * 8, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(blk_off.p, h_off.data(), ((uint64_t)nb + 1) * 8, hipMemcpyHostToDevice, s));
  SPAM_HIP_TRY(hipMemcpyAsync(pe.p, h_pe.data(), ((uint64_t)nb + 1) * 8, hipMemcpyHostToDevice, s));
  rg_fill_kernel<T><<<(unsigned)((g->padded + 255) / 256), 256, 0, s>>>(
      keys_out.as<uint64_t>(), src_out.as<uint32_t>(), start.as<uint64_t>(), pe.as<uint64_t>(), blk_off.as<uint64_t>(),
      grp_row.as<uint32_t>(), grp_sub.as<uint32_t>(), nb, ng, (uint32_t)abits, (const T*)a->vals, stream.as<unsigned char>(), vsrc.as<uint32_t>(), g->padded);
  SPAM_CHECK_LAUNCH();
  SPAM_HIP_TRY(hipStreamSynchronize(s));  // host vectors and temporaries go out of scope
  g->stream = stream.as<unsigned char>(), stream.release();
  g->blk_off = blk_off.as<uint64_t>(), blk_off.release();
  g->pe = pe.as<uint64_t>(), pe.release();
  g->grp_row = grp_row.as<uint32_t>(), grp_row.release();
  g->grp_sub = grp_sub.as<uint32_t>(), grp_sub.release();
  g->sub_head = sub_head.as<uint32_t>(), sub_head.release();
  g->rank_rows = rank_rows.as<uint32_t>(), rank_rows.release();
  g->vsrc = vsrc.as<uint32_t>(), vsrc.release();
  RgArgs<T> none{};
  SPAM_TRY(rg_dispatch<T>(g, &none, s, &g->grid));
  guard.g = nullptr;
  *out = g;
  return SPAM_OK;
}

template <class T>
int rg_run(const RgPlan* g, const T* xh, uint32_t S, const T* x, T* partial, uint32_t nl, uint32_t* queue, hipStream_t s) {
  RgArgs<T> P;
  P.blk_off = g->blk_off, P.grp_row = g->grp_row, P.grp_sub = g->grp_sub, P.sub_head = g->sub_head, P.stream = g->stream;
  for (int p = 0; p < SPMV_NSLAB; ++p) P.base[p] = xh + (uint64_t)p * S;
  P.base[SPMV_NSLAB] = x;
  P.partial = partial, P.queue = queue, P.ng = g->ng, P.nl = nl, P.abits = g->abits;
  return rg_dispatch<T>(g, &P, s, nullptr);
}

template <class T>
int rg_refresh_values(const spam_csr* a, RgPlan* g, hipStream_t s) {
  if (g->padded == 0) return SPAM_OK;
  rg_refresh_kernel<T><<<(unsigned)((g->padded + 255) / 256), 256, 0, s>>>(g->pe, g->blk_off, g->ng * SPMV_NPART, g->ng, g->vsrc,
                                                                          (const T*)a->vals, g->stream, g->padded);
  SPAM_CHECK_LAUNCH();
  return SPAM_OK;
}

#define SPAM_RG_INSTANTIATE(T)                                                                                              \
  template int rg_build<T>(const spam_csr*, const uint8_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t,            \
                           const RgKnobs&, hipStream_t, RgPlan**);                                                           \
  template int rg_run<T>(const RgPlan*, const T*, uint32_t, const T*, T*, uint32_t, uint32_t*, hipStream_t);                 \
  template int rg_refresh_values<T>(const spam_csr*, RgPlan*, hipStream_t);
SPAM_RG_INSTANTIATE(double)
SPAM_RG_INSTANTIATE(float)
SPAM_RG_INSTANTIATE(i64w)
#undef SPAM_RG_INSTANTIATE

}  // namespace spam
