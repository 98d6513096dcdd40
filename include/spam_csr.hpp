// spam_csr.hpp — C++ host-side mirror of the reference's `spam_csr::CsrMatrix<T, IS_SORTED>` for the
// SpMV / SpGEMM path, over the C ABI of libspam_hip.so (spam_hip.h).  Header-only; link with -lspam_hip.
//
// The reference is Rust and the build image has no Rust toolchain, so this is the compiled-language host
// side above the ABI (INTEGRATION.md holds the Rust shim a maintainer would add instead).  Names and
// behaviour follow the Rust type: struct fields spam_csr/src/lib.rs:25-32, Matrix trait
// spam_matrix/src/lib.rs:15-27, identity lib.rs:177-185, mul_hash mul_hash.rs:13-36, Mul lib.rs:292-297.
// Where the reference panics this throws spam::Panic.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "spam_hip.h"

namespace spam {

struct Panic : std::runtime_error {
  int code;
  Panic(int c, const char* msg) : std::runtime_error(msg), code(c) {}
};
inline void check(int status) {
  if (status != SPAM_OK) throw Panic(status, spam_last_error());
}

template <class T, bool IS_SORTED>
class CsrMatrix {
  static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value || std::is_same<T, int64_t>::value,
                "the HIP back end serves f32 / f64 / Wrapping<i64>; other T keep the reference's CPU path");

 public:
  // CsrMatrix { rows, cols, vals, indices, offsets }  (lib.rs:26-32)
  CsrMatrix(uint64_t rows, uint64_t cols, std::vector<T> vals, std::vector<uint64_t> indices,
            std::vector<uint64_t> offsets)
      : rows_(rows), cols_(cols), vals_(std::move(vals)), indices_(std::move(indices)), offsets_(std::move(offsets)) {
    if (rows == 0 || cols == 0) throw Panic(SPAM_E_INVALID, "rows and cols are NonZeroUsize");
  }
  CsrMatrix(const CsrMatrix&) = delete;
  CsrMatrix& operator=(const CsrMatrix&) = delete;
  CsrMatrix(CsrMatrix&& o) noexcept { *this = std::move(o); }
  CsrMatrix& operator=(CsrMatrix&& o) noexcept {
    release();
    rows_ = o.rows_, cols_ = o.cols_;
    vals_ = std::move(o.vals_), indices_ = std::move(o.indices_), offsets_ = std::move(o.offsets_);
    dev_ = o.dev_;
    o.dev_ = nullptr;
    return *this;
  }
  ~CsrMatrix() { release(); }

  // Matrix::new / new_square / identity  (lib.rs:162-185)
  static CsrMatrix new_(uint64_t rows, uint64_t cols) {
    return CsrMatrix(rows, cols, {}, {}, std::vector<uint64_t>(rows + 1, 0));
  }
  static CsrMatrix new_square(uint64_t n) { return new_(n, n); }
  static CsrMatrix identity(uint64_t n) {
    std::vector<uint64_t> idx(n), off(n + 1);
    for (uint64_t i = 0; i < n; ++i) idx[i] = i;
    for (uint64_t i = 0; i <= n; ++i) off[i] = i;
    return CsrMatrix(n, n, std::vector<T>(n, T(1)), std::move(idx), std::move(off));
  }

  uint64_t rows() const { return rows_; }
  uint64_t cols() const { return cols_; }
  uint64_t nnz() const { return indices_.size(); }
  const std::vector<T>& vals() const { return vals_; }
  const std::vector<uint64_t>& indices() const { return indices_; }
  const std::vector<uint64_t>& offsets() const { return offsets_; }

  // Matrix::invariants (lib.rs:152-160): 1 and 2 on the host, 3-7 on the device
  bool invariants() const {
    if (indices_.size() != vals_.size() || offsets_.size() != rows_ + 1) return false;
    int ok = 0;
    try {
      check(spam_csr_invariants(device(), &ok));
    } catch (const Panic& p) {
      if (p.code == SPAM_E_INVALID || p.code == SPAM_E_INDEX_RANGE) return false;
      throw;
    }
    return ok != 0;
  }

  // pub fn mul_hash<const B1: bool, const B2: bool>(&self, rhs: &CsrMatrix<T, B1>) -> CsrMatrix<T, B2>
  template <bool B1, bool B2>
  CsrMatrix<T, B2> mul_hash(const CsrMatrix<T, B1>& rhs) const {
    spam_spgemm_plan* plan = nullptr;
    uint64_t nnz = 0;
    check(spam_spgemm_symbolic(device(), rhs.device(), &plan, &nnz));
    std::vector<uint64_t> offsets(rows_ + 1), indices(nnz);  // Vec::with_capacity(nnz) (mul_hash.rs:119)
    std::vector<T> vals(nnz);
    int st;
    if constexpr (std::is_same<T, double>::value)
      st = spam_spgemm_numeric_f64(plan, B2 ? 1 : 0, offsets.data(), indices.data(), vals.data());
    else if constexpr (std::is_same<T, int64_t>::value)
      st = spam_spgemm_numeric_i64(plan, B2 ? 1 : 0, offsets.data(), indices.data(), vals.data());
    else
      st = spam_spgemm_numeric_f32(plan, B2 ? 1 : 0, offsets.data(), indices.data(), vals.data());
    spam_spgemm_plan_free(plan);
    check(st);
    return CsrMatrix<T, B2>(rows_, rhs.cols(), std::move(vals), std::move(indices), std::move(offsets));
  }

  // impl Mul for &CsrMatrix<T, B>  ->  CsrMatrix<T, false>  (lib.rs:292-297)
  CsrMatrix<T, false> operator*(const CsrMatrix& rhs) const { return mul_hash<IS_SORTED, false>(rhs); }

  // impl Add / Sub for CsrMatrix (lib.rs:276-290) through apply_elementwise (lib.rs:83-149): sorted-union rows,
  // results equal to zero are kept
  CsrMatrix<T, true> operator+(const CsrMatrix& rhs) const { return ewise(rhs, 0); }
  CsrMatrix<T, true> operator-(const CsrMatrix& rhs) const { return ewise(rhs, 1); }

  // the SpMV analysis on the uploaded copy (spam_spmv_prepare): worth it when the matrix is multiplied many times
  void spmv_prepare() const { check(spam_spmv_prepare(const_cast<spam_csr*>(device()), nullptr)); }

  // y = A x (absent upstream; mul_hash with an n x 1 right-hand side)
  std::vector<T> spmv(const std::vector<T>& x) const {
    if (x.size() != cols_) throw Panic(SPAM_E_INVALID, "x.len() != cols");
    std::vector<T> y(rows_);
    if constexpr (std::is_same<T, double>::value)
      check(spam_spmv_f64(device(), x.data(), y.data()));
    else if constexpr (std::is_same<T, int64_t>::value)
      check(spam_spmv_i64(device(), x.data(), y.data()));
    else
      check(spam_spmv_f32(device(), x.data(), y.data()));
    return y;
  }

  // the uploaded copy (created on first use, owned by this object)
  const spam_csr* device() const {
    if (!dev_) {
      if constexpr (std::is_same<T, double>::value)
        check(spam_csr_upload_f64(rows_, cols_, indices_.size(), offsets_.data(), indices_.data(), vals_.data(),
                                  IS_SORTED, &dev_));
      else if constexpr (std::is_same<T, int64_t>::value)
        check(spam_csr_upload_i64(rows_, cols_, indices_.size(), offsets_.data(), indices_.data(), vals_.data(),
                                  IS_SORTED, &dev_));
      else
        check(spam_csr_upload_f32(rows_, cols_, indices_.size(), offsets_.data(), indices_.data(), vals_.data(),
                                  IS_SORTED, &dev_));
    }
    return dev_;
  }

 private:
  CsrMatrix<T, true> ewise(const CsrMatrix& rhs, int subtract) const {
    spam_csr* c = nullptr;
    check(spam_csr_add(device(), rhs.device(), subtract, &c));
    uint64_t rows = 0, cols = 0, nnz = 0;
    check(spam_csr_dims(c, &rows, &cols, &nnz, nullptr, nullptr));
    std::vector<uint64_t> offsets(rows + 1), indices(nnz);
    std::vector<T> vals(nnz);
    int st;
    if constexpr (std::is_same<T, double>::value)
      st = spam_csr_download_f64(c, offsets.data(), indices.data(), vals.data());
    else if constexpr (std::is_same<T, int64_t>::value)
      st = spam_csr_download_i64(c, offsets.data(), indices.data(), vals.data());
    else
      st = spam_csr_download_f32(c, offsets.data(), indices.data(), vals.data());
    spam_csr_free(c);
    check(st);
    return CsrMatrix<T, true>(rows, cols, std::move(vals), std::move(indices), std::move(offsets));
  }
  void release() {
    if (dev_) spam_csr_free(dev_);
    dev_ = nullptr;
  }
  uint64_t rows_ = 0, cols_ = 0;
  std::vector<T> vals_;
  std::vector<uint64_t> indices_, offsets_;
  mutable spam_csr* dev_ = nullptr;
};

}  // namespace spam
