/* spam_hip.h — C ABI of libspam_hip.so: the MI355X (gfx950) back end for the spam_csr hot path.
 *
 * What this replaces.  The reference (sledgehammervampire/sparse_matrix, Rust) has no FFI or plug-in
 * interface; its one product is `CsrMatrix::mul_hash::<B1, B2>` (spam_csr/src/mul_hash.rs:13-36),
 * reached through `impl Mul for &CsrMatrix` (spam_csr/src/lib.rs:292-297) and through the method-name
 * macros `gen_bench_mul!($f)` / `gen_mul_main!($f)` (spam_csr/src/lib.rs:386-467).  A backend therefore
 * lives in a new module next to `mod mul_hash;` (lib.rs:21) and binds the entry points below from an
 * `extern "C"` block (INTEGRATION.md shows the Rust shim).  Every entry point cites the reference code
 * it stands in for.  Plain pointers and sizes only; all HIP runtime use is inside the library.
 *
 * Host layout == the reference's: `offsets: Vec<usize>`, `indices: Vec<usize>`, `vals: Vec<T>`
 * (spam_csr/src/lib.rs:25-32), usize == uint64_t.  Device layout: uint64 row_ptr, uint32 col_idx
 * (the reference already truncates column keys to u32, mul_hash.rs:12,92,157), T vals.
 *
 * Errors: the reference panics (checked_add unwrap mul_hash.rs:47-48 / lib.rs:270; slice OOB on an
 * inner-dimension mismatch mul_hash.rs:46; assert_eq! mul_hash.rs:190).  Here every function returns a
 * status and records a message for spam_last_error(); the Rust shim turns non-zero into panic!().
 * Unlike the reference, A.cols == B.rows is checked up front.
 *
 * Threading: a handle may be used by several host threads at once for SpMV / SpGEMM / downloads (the lazily
 * built SpMV tile map is created under a per-handle lock, and the launches of one SpMV — which share the plan's
 * scratch buffers — are enqueued as a unit under a second one); spam_spmv_prepare, spam_csr_values_changed,
 * spam_csr_sort_rows and spam_csr_free need exclusive use of the handle.  The stream selection is process-global and
 * the device allocator recycles blocks without stream ordering: use ONE stream per process (spam_hip_set_stream once,
 * before any other call).  One process drives one GPU (one rank per GPU under RCCL).
 */
#ifndef SPAM_HIP_H
#define SPAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPAM_OK 0
#define SPAM_E_INVALID 1     /* null/ill-formed argument, dimension mismatch (ref: OOB panic, mul_hash.rs:46) */
#define SPAM_E_INDEX_RANGE 2 /* cols >= 2^32-1: column keys must fit u32 and != u32::MAX (mul_hash.rs:12, set.rs:110) */
#define SPAM_E_OOM 3         /* device or host allocation failed */
#define SPAM_E_HIP 4         /* HIP runtime error (message has the hipError string) */
#define SPAM_E_OVERFLOW 5    /* checked_add overflow (mul_hash.rs:47-48, lib.rs:270) */
#define SPAM_E_UNSUPPORTED 6 /* valid request this build cannot serve (message says why) */
#define SPAM_E_DTYPE 7       /* entry point of one value type called on a handle of another */
#define SPAM_E_COMM 8        /* RCCL error in libspam_hip_rccl.so (spam_hip_rccl.h; message has ncclGetErrorString) */

#define SPAM_F32 0
#define SPAM_F64 1
#define SPAM_I64 2 /* Wrapping<i64>: two's-complement wrap-around, the arithmetic of the reference's own hot-path test
                    * (Wrapping<i8>, spam_csr/src/tests.rs:356-371) and of its MatrixType::Integer bench files (parsed
                    * as i64, spam_csr/src/lib.rs:421-424).  Integer sums do not depend on the order of the additions, so
                    * the GPU result equals the reference BIT FOR BIT in values as well as in structure. */

typedef struct spam_csr spam_csr;                 /* device-resident CsrMatrix<T, IS_SORTED> */
typedef struct spam_spgemm_plan spam_spgemm_plan; /* state between symbolic and numeric (mul_hash.rs:22,28) */

/* Measured facts about one SpGEMM call; all counts exact. */
typedef struct spam_spgemm_stats {
  uint64_t rows, inner, cols;      /* C is rows x cols, inner = A.cols = B.rows */
  uint64_t nnz_a, nnz_b, nnz_c;
  uint64_t flops_f;                /* F = sum of intermediate products (row_nz before symbolic, mul_hash.rs:39-50) */
  uint64_t rows_sym_bin[8];        /* rows per symbolic bin (by F)   */
  uint64_t rows_num_bin[8];        /* rows per numeric bin (by nnz)  */
  uint64_t heavy_items;            /* rows on the heavy numeric path (more than 4096 entries of C) */
  uint64_t heavy_dense_groups, heavy_hash_groups; /* their (row, column group) work items: rank-addressed accumulator /
                                                   * (none since round 2: the per-group hash table is gone) */
  uint64_t heavy_dense_nnz, heavy_hash_nnz;       /* entries of C produced by each kind */
  float ms_flops, ms_symbolic, ms_alloc, ms_numeric, ms_total; /* device time, HIP events on the library stream */
} spam_spgemm_stats;

/* ---- runtime ------------------------------------------------------------------------------- */

/* Select the GPU this process drives (hipSetDevice) and check it is gfx950.  device = LOCAL_RANK. */
int spam_hip_init(int device);
/* Use an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream) for all launches; NULL = default stream. */
int spam_hip_set_stream(void* hip_stream);
/* Block until everything queued on the library stream has finished. */
int spam_hip_synchronize(void);
/* The library keeps freed device blocks for reuse (a 100 GB C costs seconds of hipMalloc otherwise);
 * this hands all cached blocks back to the driver. */
int spam_hip_trim(void);
/* Measurement aid: effective GB/s (bytes read + bytes written) of a float4 copy kernel over `bytes` bytes, `reps`
 * launches — the streaming ceiling of this device, measured in the caller's process (bench.py prints it beside the
 * roofline fractions; MI355X_MICROARCH.md quotes 6.29 TB/s). */
int spam_hip_copy_gbps(uint64_t bytes, int reps, double* gbps_out);
/* Message of the last failing call on this thread ("" if none).  Never NULL. */
const char* spam_last_error(void);
/* Library/ABI version: major*1000 + minor. */
int spam_version(void);

/* ---- CsrMatrix <-> device ------------------------------------------------------------------ */

/* Upload a host CsrMatrix (struct fields, spam_csr/src/lib.rs:25-32).  Narrows indices to u32 on the
 * device; fails with SPAM_E_INDEX_RANGE if cols >= 2^32-1, SPAM_E_INVALID if the arrays break
 * invariants 2,3,4,5,7 (lib.rs:51-67,79-81).  is_sorted = the const generic IS_SORTED. */
int spam_csr_upload_f64(uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* offsets,
                        const uint64_t* indices, const double* vals, int is_sorted, spam_csr** out);
int spam_csr_upload_f32(uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* offsets,
                        const uint64_t* indices, const float* vals, int is_sorted, spam_csr** out);
int spam_csr_upload_i64(uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* offsets,
                        const uint64_t* indices, const int64_t* vals, int is_sorted, spam_csr** out);

/* Wrap arrays already in HBM (no copy, caller keeps ownership and must outlive the handle):
 * d_row_ptr uint64[rows+1], d_col_idx uint32[nnz], d_vals T[nnz]; all 16-byte aligned. */
int spam_csr_from_device(int dtype, uint64_t rows, uint64_t cols, uint64_t nnz, const uint64_t* d_row_ptr,
                         const uint32_t* d_col_idx, const void* d_vals, int is_sorted, spam_csr** out);

/* CSR from (row, col)-sorted triples already in HBM — the device-side `From<DokMatrix>` (BTreeMap order,
 * spam_csr/src/lib.rs:315-334): builds row_ptr (owned by the handle), borrows d_col_idx / d_vals.
 * SPAM_E_INVALID if the row indices are not non-decreasing or reach `rows`. */
int spam_csr_from_sorted_coo(int dtype, uint64_t rows, uint64_t cols, uint64_t nnz, const uint32_t* d_row_idx,
                             const uint32_t* d_col_idx, const void* d_vals, int is_sorted, spam_csr** out);

/* A view of rows [row_lo, row_hi) of `a` as its own matrix (shares col_idx/vals with `a`, which must
 * outlive it).  Used to hand each GPU its row block of A (rows_to_threads with tnum = #GPUs). */
int spam_csr_row_block(const spam_csr* a, uint64_t row_lo, uint64_t row_hi, spam_csr** out);

int spam_csr_dims(const spam_csr* m, uint64_t* rows, uint64_t* cols, uint64_t* nnz, int* dtype, int* is_sorted);
/* Raw device pointers (for RCCL all-gather of B's arrays by the caller, or zero-copy torch views). */
int spam_csr_device_ptrs(const spam_csr* m, const uint64_t** d_row_ptr, const uint32_t** d_col_idx,
                         const void** d_vals);
/* Copy back into caller-allocated host Vecs (offsets[rows+1], indices[nnz] widened to usize, vals[nnz]). */
int spam_csr_download_f64(const spam_csr* m, uint64_t* offsets, uint64_t* indices, double* vals);
int spam_csr_download_f32(const spam_csr* m, uint64_t* offsets, uint64_t* indices, float* vals);
int spam_csr_download_i64(const spam_csr* m, uint64_t* offsets, uint64_t* indices, int64_t* vals);
/* Sort every row by column in place: CsrMatrix<T, false> -> CsrMatrix<T, true> (what `From<DokMatrix>` guarantees,
 * spam_csr/src/lib.rs:315-334).  Only for handles that own their arrays (uploads, SpGEMM results). */
int spam_csr_sort_rows(spam_csr* m);
/* CsrMatrix::invariants() 1-7 evaluated on the device (spam_csr/src/lib.rs:47-81,152-160).  *ok = 0/1. */
int spam_csr_invariants(const spam_csr* m, int* ok);
void spam_csr_free(spam_csr* m);

/* C = A + B (subtract = 0) or A - B (subtract = 1): `impl Add / Sub for CsrMatrix` through apply_elementwise
 * (spam_csr/src/lib.rs:83-149, 276-290).  Rows of C are the sorted union of the rows of A and B; entries whose
 * result is zero are kept, as in the reference.  Unsorted operands are sorted in private copies first (the
 * reference's HashMap path leaves their order unspecified).  Dimensions must match (assert_eq!, lib.rs:87-91). */
int spam_csr_add(const spam_csr* a, const spam_csr* b, int subtract, spam_csr** out);

/* ---- SpMV: y = A x --------------------------------------------------------------------------
 * Absent from the reference; defined as mul_hash with an n x 1 right-hand side (SURVEY.md §3.5,
 * mul_hash.rs:145-163): rows without entries give 0.  x has cols(A) entries, y rows(A). */
int spam_spmv_f64(const spam_csr* a, const double* x_host, double* y_host); /* host slices (&[T] -> Vec<T>) */
int spam_spmv_f32(const spam_csr* a, const float* x_host, float* y_host);
int spam_spmv_i64(const spam_csr* a, const int64_t* x_host, int64_t* y_host);
/* Device-resident vectors, asynchronous.  d_x and d_y must not overlap.  On a handle that was prepared, or has
 * multiplied once, the call only enqueues kernels on the library's stream (no allocation, synchronisation or host
 * read-back): it may be recorded into a HIP graph by stream capture and replayed (tests/test_spmv_sliced_gpu.py). */
int spam_spmv_dev(const spam_csr* a, const void* d_x, void* d_y);
/* The analysis step (CSR-adaptive's row binning; here: row -> tile map, hot-column table, XCD-sliced structure), done
 * ONCE per matrix and never inside spam_spmv_*: reads the SPAM_SPMV_* environment knobs, synchronises with the host,
 * and — from a million entries on — times three SpMVs of the plain path on d_x (NULL: a vector of ones); unless that
 * already streams the matrix at 4.5 TB/s (banded matrices do), it builds the XCD-sliced structure where the matrix
 * qualifies, times it the same way and keeps the faster (the sliced one must win by 3 %).  The choice fixes the summation order of later SpMVs on
 * this handle.  A handle that was never prepared multiplies through the plain tile path without hot-column table: its
 * first SpMV only adds one small kernel (no analysis, no timing), so a one-shot y = A x costs one pass over A.
 * Prepare again after changing the knobs; not to be called while SpMVs on this handle are being queued elsewhere. */
int spam_spmv_prepare(spam_csr* a, const void* d_x);
/* The SpMV plan of a prepared handle may hold private copies of the matrix VALUES (the XCD-sliced structure stores the
 * long rows again, slab by slab).  After updating vals in place — allowed for arrays wrapped by spam_csr_from_device /
 * from_sorted_coo — call this to gather them again (no new analysis, no new timing of the paths).  A row-group plan
 * keeps no map from its value positions back to the matrix (4 bytes per entry) unless SPAM_SPMV_RG_VSRC=1: the FIRST
 * call builds that plan again with the map (once: what spam_spmv_prepare spent on it, host-synchronised), later calls
 * are asynchronous and take a few ms for 160 M entries.  The
 * STRUCTURE (row_ptr, col_idx) of a handle must never change: wrap a new handle instead. */
int spam_csr_values_changed(spam_csr* a);
/* What the SpMV analysis of this matrix decided (valid after spam_spmv_prepare): number of columns in the
 * L2-resident hot table (0 = disabled), the share of the entries they cover, and the number of nnz tiles. */
int spam_spmv_plan_info(const spam_csr* a, uint64_t* hot_columns, double* hot_cover, uint64_t* tiles);
/* XCD-sliced mode (valid after spam_spmv_prepare): the analysis deals the most referenced columns over 8 slabs, one
 * x-table per XCD's L2, and stores the rows of >= long_min entries slab by slab — in row-group form (below; a candidate
 * from 3 M entries on, whatever the size of x) or, with SPAM_SPMV_RG=0, as 8 slab CSRs + 8 partial y vectors added in
 * slab order (a candidate when x is larger than one L2 and eight tables cover clearly more gathers than one) — and the
 * shorter rows in a residual CSR; spam_spmv_prepare times both paths and keeps the faster one.  *active = 1 when the
 * sliced path won; the other outputs describe it (rows / entries in the slabs, table slots per XCD, slots also held
 * in LDS, share of the gathers covered by the 8 tables / by one table, the two timings in ms).  Environment:
 * SPAM_SPMV_SLICED = 0 never, 1 decide (default), 2 always. */
int spam_spmv_sliced_info(const spam_csr* a, int* active, uint64_t* long_rows, uint64_t* nnz_long, uint64_t* table_slots,
                          uint64_t* lds_slots, double* cover_table, double* cover_single, double* ms_plain,
                          double* ms_sliced);
/* The long rows of a sliced plan are kept in ROW-GROUP form unless SPAM_SPMV_RG=0 (spam_spmv_rg.hip): per part and
 * group of up to 8191 rows one block (16 k .. 200 k entries, following the size of the matrix; SPAM_SPMV_RG_EBLK) whose
 * entries are stored in table-slot order and whose y accumulators live in LDS.  *active = 1 when the handle's plan has that form; row groups, non-empty (part, group) blocks, accumulators
 * (rows that would serialise LDS atomics get several), bytes of the entry stream one SpMV reads, bytes of HBM the
 * form holds.  In this form f32 / f64 sums are accumulated by LDS atomics: results agree with the reference to
 * rounding (1e-10 / 1e-5 bars) but are not bit-identical from run to run; i64 results are exact. */
int spam_spmv_rg_info(const spam_csr* a, int* active, uint64_t* row_groups, uint64_t* blocks, uint64_t* accumulators,
                      uint64_t* stream_bytes, uint64_t* plan_bytes);

/* ---- SpGEMM: C = A B, two phases like mul_hash ----------------------------------------------
 * spam_spgemm_symbolic  = rows_to_threads' flop count (mul_hash.rs:39-50) + mul_hash_symbolic (:66-103)
 *                         + checked_inclusive_scan (lib.rs:267-274); *nnz_out = nnz(C), so the caller can
 *                         Vec::with_capacity(nnz) exactly as mul_hash.rs:119 does.
 * spam_spgemm_numeric_* = mul_hash_numeric::<B1, B2> (:106-201); sort_rows is the const generic B2
 *                         (rows sorted by column).  With sort_rows = 0 each row is emitted in this
 *                         library's own order — rows of up to 4096 entries in the slot order of a table filled
 *                         in parallel, longer rows in column order: the same SET of (col, value) per row as the
 *                         reference, not its linprobe slot order.  sort_rows = 2 (SPAM_ORDER_REFERENCE)
 *                         reproduces the reference's unsorted output exactly — linprobe slot order,
 *                         map.rs:59-63 (rows above 4096 entries build their table in HBM: a verification
 *                         path, several times slower).  Explicit zeros are kept (structure = symbolic
 *                         pattern), as in the reference. */
#define SPAM_ORDER_TABLE 0
#define SPAM_ORDER_SORTED 1
#define SPAM_ORDER_REFERENCE 2
int spam_spgemm_symbolic(const spam_csr* a, const spam_csr* b, spam_spgemm_plan** plan, uint64_t* nnz_out);
int spam_spgemm_numeric_f64(spam_spgemm_plan* plan, int sort_rows, uint64_t* offsets_out /* rows+1 */,
                            uint64_t* indices_out /* nnz */, double* vals_out /* nnz */);
int spam_spgemm_numeric_f32(spam_spgemm_plan* plan, int sort_rows, uint64_t* offsets_out, uint64_t* indices_out,
                            float* vals_out);
int spam_spgemm_numeric_i64(spam_spgemm_plan* plan, int sort_rows, uint64_t* offsets_out, uint64_t* indices_out,
                            int64_t* vals_out);
/* Same, but C stays in HBM as a new handle (free with spam_csr_free). */
int spam_spgemm_numeric_dev(spam_spgemm_plan* plan, int sort_rows, spam_csr** c_out);
int spam_spgemm_plan_stats(const spam_spgemm_plan* plan, spam_spgemm_stats* stats);
void spam_spgemm_plan_free(spam_spgemm_plan* plan);

/* One call, device-resident: symbolic + allocate + numeric.  stats may be NULL. */
int spam_spgemm_dev(const spam_csr* a, const spam_csr* b, int sort_rows, spam_csr** c_out,
                    spam_spgemm_stats* stats);

/* One rank's share of the row-sharded product C = A B on `world` GPUs, in one call: rows_to_threads' partition
 * (mul_hash.rs:38-64) with tnum = world on F_i + row_overhead * [F_i > 0] (computed on the device; every rank gets the
 * same offsets), a zero-copy view of A's rows [*row_lo, *row_hi) and the two-phase SpGEMM of that block with all of B.
 * a_full / b_full are this rank's complete copies of A and B (for C = A A the same handle; B arrives through the
 * all-gather of the ranks' row shards, which is the HOST's step: exchange the three arrays of spam_csr_device_ptrs
 * with RCCL — INTEGRATION.md shows the grouped ncclSend / ncclRecv sequence — and wrap the result with
 * spam_csr_from_device).  *c_block = the block of C (NULL when the block has no rows); C stays row-sharded.
 * row_overhead: SPAM_ROW_OVERHEAD_AUTO (below) unless the caller knows better. */
int spam_spgemm_rank_step(const spam_csr* a_full, const spam_csr* b_full, uint64_t rank, uint64_t world,
                          uint64_t row_overhead, int sort_rows, spam_csr** c_block, uint64_t* row_lo, uint64_t* row_hi,
                          spam_spgemm_stats* stats);

/* row_overhead = SPAM_ROW_OVERHEAD_AUTO lets the library pick the per-row term from the width of B (32 products per
 * 8192-column chunk of B: a non-empty row of the heavy path costs work per chunk it is cut along; measured on MI355X at
 * R-MAT scale 20 and 22, tools/partition_balance.py).  0 is the reference's rule on F alone.  spam_row_overhead_auto
 * returns the value AUTO stands for. */
#define SPAM_ROW_OVERHEAD_AUTO 0xFFFFFFFFFFFFFFFFull
uint64_t spam_row_overhead_auto(const spam_csr* b);

/* rows_to_threads (mul_hash.rs:38-64) with tnum = nparts: per-row intermediate-product counts and the
 * flop-balanced contiguous row ranges rows_offset[0..nparts].  row_flops_out may be NULL. */
int spam_rows_to_parts(const spam_csr* a, const spam_csr* b, uint64_t nparts, uint64_t* row_flops_out /* rows */,
                       uint64_t* rows_offset_out /* nparts+1 */);
/* The same partition rule on F_i + row_overhead * [F_i > 0]: on the GPU a block of many light rows runs longer
 * than a block of few heavy rows with equal F (measured: a non-empty row costs about 4000 products; 2300 with the round-1 kernels). */
int spam_rows_to_parts_weighted(const spam_csr* a, const spam_csr* b, uint64_t nparts, uint64_t row_overhead,
                                uint64_t* rows_offset_out /* nparts+1 */);

/* The two halves of rows_to_threads, device-resident, for hosts that spread the flop count over the ranks (the
 * multi-GPU step of libspam_hip_rccl.so does): (1) mul_hash.rs:39-50 for rows [row_lo, row_hi) of A only — reads A's row
 * range and B's row_ptr, writes d_flops_out[0 .. row_hi - row_lo) in HBM; SPAM_E_OVERFLOW where checked_add would panic
 * (partial sums of 2^60 and more are refused, so that the sum over up to 8 ranks' shares still fits);
 * (2) mul_hash.rs:51-62 on a device vector of per-row counts (+ row_overhead per non-empty row, see above). */
int spam_row_flops_dev(const spam_csr* a, const spam_csr* b, uint64_t row_lo, uint64_t row_hi, uint64_t* d_flops_out);
int spam_rows_to_parts_from_flops_dev(const uint64_t* d_row_flops, uint64_t rows, uint64_t nparts, uint64_t row_overhead,
                                      uint64_t* rows_offset_out /* nparts+1, host */);

#ifdef __cplusplus
}
#endif
#endif /* SPAM_HIP_H */
