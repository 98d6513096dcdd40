/* A plain-C host driving one rank of the multi-GPU product through the C ABI alone (include/spam_hip.h +
 * include/spam_hip_rccl.h) — the call sequence INTEGRATION.md §5 gives for the Rust host, compiled with gcc as C to show
 * that the boundary needs neither C++ nor an NCCL binding on the host side.  Run by tests/test_rccl_cabi_gpu.py in a
 * communicator of ONE rank (a 1-GPU box; RCCL refuses two ranks on one device): every ncclSend meets its ncclRecv on
 * the same device, so the grouped exchange, the row_ptr fix-up, the event that holds back readers of the values and the
 * sharded flop count all execute.  Usage: c_rccl_host [rank world idfile]  (rank 0 writes the 128-byte id to idfile,
 * the other ranks read it: that is the whole bootstrap a launcher has to provide). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spam_hip_rccl.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int s_ = (call);                                                                     \
    if (s_ != SPAM_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, s_, spam_last_error());                   \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

int main(int argc, char** argv) {
  const int rank = argc > 2 ? atoi(argv[1]) : 0, world = argc > 2 ? atoi(argv[2]) : 1;
  const char* idfile = argc > 3 ? argv[3] : NULL;
  CHECK(spam_hip_init(argc > 4 ? atoi(argv[4]) : rank));

  /* bootstrap: 128 opaque bytes from rank 0 to everyone */
  unsigned char id[SPAM_RCCL_ID_BYTES];
  if (rank == 0) {
    CHECK(spam_rccl_unique_id(id));
    if (idfile) {
      FILE* f = fopen(idfile, "wb");
      if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return fprintf(stderr, "cannot write %s\n", idfile), 1;
      fclose(f);
    }
  } else {
    FILE* f = fopen(idfile, "rb");
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) return fprintf(stderr, "cannot read %s\n", idfile), 1;
    fclose(f);
  }
  spam_comm* comm = NULL;
  CHECK(spam_rccl_comm_create(id, rank, world, &comm));

  /* A = a small matrix with a pattern every rank can rebuild: row i holds columns (i*j + 1) % n for j < 1 + i % 7,
   * duplicates skipped, unsorted on purpose (IS_SORTED = false), Wrapping<i64> values so that the check is exact */
  const uint64_t n = 600;
  uint64_t* off = calloc(n + 1, 8);
  uint64_t* idx = malloc(n * 8 * 8);
  int64_t* val = malloc(n * 8 * 8);
  uint64_t nnz = 0;
  for (uint64_t i = 0; i < n; ++i) {
    for (uint64_t j = 0; j < 1 + i % 7; ++j) {
      const uint64_t c = (i * j + 1) % n;
      int dup = 0;
      for (uint64_t p = off[i]; p < nnz; ++p) dup |= idx[p] == c;
      if (dup) continue;
      idx[nnz] = c, val[nnz] = (int64_t)(i * 31 + j * 7) - 9000, ++nnz;
    }
    off[i + 1] = nnz;
  }
  /* this rank's row shard (equal row blocks), uploaded with shard-local offsets exactly like any CsrMatrix */
  uint64_t* row_offsets = malloc((world + 1) * 8);
  for (int r = 0; r <= world; ++r) row_offsets[r] = n * (uint64_t)r / (uint64_t)world;
  const uint64_t lo = row_offsets[rank], hi = row_offsets[rank + 1];
  uint64_t* soff = malloc((hi - lo + 1) * 8);
  for (uint64_t i = lo; i <= hi; ++i) soff[i - lo] = off[i] - off[lo];
  spam_csr *shard = NULL, *b_full = NULL, *c_blk = NULL;
  CHECK(spam_csr_upload_i64(hi - lo, n, off[hi] - off[lo], soff, idx + off[lo], val + off[lo], 0, &shard));

  /* the exchange: B's three arrays from every rank (vals last, still in flight when this returns) */
  CHECK(spam_csr_allgather_rccl(comm, shard, row_offsets, &b_full));
  uint64_t recv = 0, sent = 0, r_, c_, z_;
  CHECK(spam_rccl_last_exchange(comm, &recv, &sent));
  CHECK(spam_csr_dims(b_full, &r_, &c_, &z_, NULL, NULL));
  if (r_ != n || c_ != n || z_ != nnz) return fprintf(stderr, "gathered dims %llu x %llu, %llu entries\n", (unsigned long long)r_, (unsigned long long)c_, (unsigned long long)z_), 1;

  /* partition (flop count sharded + all-gathered) + this rank's block of C = A A, sorted rows (B2 = true) */
  uint64_t blo = 0, bhi = 0;
  spam_spgemm_stats st;
  CHECK(spam_spgemm_rank_step_rccl(comm, b_full, b_full, NULL, SPAM_ROW_OVERHEAD_AUTO, SPAM_ORDER_SORTED, &c_blk, &blo, &bhi, &st));

  /* the gathered matrix equals the original, bit for bit */
  uint64_t* goff = malloc((n + 1) * 8);
  uint64_t* gidx = malloc(nnz * 8);
  int64_t* gval = malloc(nnz * 8);
  CHECK(spam_csr_download_i64(b_full, goff, gidx, gval));
  if (memcmp(goff, off, (n + 1) * 8) || memcmp(gidx, idx, nnz * 8) || memcmp(gval, val, nnz * 8))
    return fprintf(stderr, "gathered B differs from the original\n"), 1;

  /* C's block against a dense row-by-row product in wrapping 64-bit arithmetic (rows_to_threads' block [blo, bhi)) */
  int bad = 0;
  if (c_blk) {
    uint64_t cr, cc, cz;
    CHECK(spam_csr_dims(c_blk, &cr, &cc, &cz, NULL, NULL));
    uint64_t* coff = malloc((cr + 1) * 8);
    uint64_t* cidx = malloc((cz + 1) * 8);
    int64_t* cval = malloc((cz + 1) * 8);
    CHECK(spam_csr_download_i64(c_blk, coff, cidx, cval));
    uint64_t* acc = malloc(n * 8);
    char* hit = malloc(n);
    for (uint64_t i = blo; i < bhi && !bad; ++i) {
      memset(acc, 0, n * 8), memset(hit, 0, n);
      for (uint64_t p = off[i]; p < off[i + 1]; ++p)
        for (uint64_t q = off[idx[p]]; q < off[idx[p] + 1]; ++q)
          acc[idx[q]] += (uint64_t)val[p] * (uint64_t)val[q], hit[idx[q]] = 1;
      uint64_t p = coff[i - blo];
      for (uint64_t j = 0; j < n; ++j)
        if (hit[j]) {
          if (p >= coff[i - blo + 1] || cidx[p] != j || (uint64_t)cval[p] != acc[j]) bad = 1;
          ++p;
        }
      if (p != coff[i - blo + 1]) bad = 1;
    }
    if (cr != bhi - blo) bad = 1;
  } else if (bhi > blo) {
    bad = 1;
  }
  if (bad) return fprintf(stderr, "block [%llu, %llu) of C differs from the triple loop\n", (unsigned long long)blo, (unsigned long long)bhi), 1;

  /* x for a row-sharded SpMV: every rank contributes its slice */
  uint64_t* counts = malloc(world * 8);
  for (int r = 0; r < world; ++r) counts[r] = row_offsets[r + 1] - row_offsets[r];
  /* (device buffers come from the caller in a real host; here the gathered row_ptr of B doubles as 8-byte payload) */
  const uint64_t* d_rp = NULL;
  CHECK(spam_csr_device_ptrs(b_full, &d_rp, NULL, NULL));
  spam_csr* scratch = NULL; /* an n x 1 matrix whose vals array is the receive buffer */
  uint64_t* zoff = calloc(n + 1, 8);
  for (uint64_t i = 0; i <= n; ++i) zoff[i] = i;
  uint64_t* zidx = calloc(n, 8);
  int64_t* zval = calloc(n, 8);
  CHECK(spam_csr_upload_i64(n, 1, n, zoff, zidx, zval, 1, &scratch));
  const void* d_recv = NULL;
  CHECK(spam_csr_device_ptrs(scratch, NULL, NULL, &d_recv));
  CHECK(spam_vector_allgather_rccl(comm, 8, d_rp + lo, counts, (void*)d_recv));
  CHECK(spam_hip_synchronize());
  CHECK(spam_csr_download_i64(scratch, NULL, NULL, zval));
  for (uint64_t i = 0; i < n; ++i)
    if ((uint64_t)zval[i] != off[i]) return fprintf(stderr, "vector all-gather: element %llu\n", (unsigned long long)i), 1;

  printf("c rccl host ok: rank %d of %d, B %llu entries gathered (%llu bytes received), block [%llu, %llu) of C: %llu entries, F = %llu\n",
         rank, world, (unsigned long long)nnz, (unsigned long long)recv, (unsigned long long)blo, (unsigned long long)bhi,
         (unsigned long long)st.nnz_c, (unsigned long long)st.flops_f);
  spam_csr_free(c_blk);
  spam_csr_free(scratch);
  spam_csr_free(b_full);
  spam_csr_free(shard);
  spam_rccl_comm_free(comm);
  return 0;
}
