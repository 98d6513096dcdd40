// spam_csr.hpp — C++ host-side mirror of the reference's `spam_csr::CsrMatrix<T, IS_SORTED>` for the
// SpMV / SpGEMM path, over the C ABI of libspam_hip.so (spam_hip.h).  Header-only; link with -lspam_hip.
//
// The reference is Rust and the build image has no Rust toolchain, so this is the compiled-language host
// side above the ABI (INTEGRATION.md holds the Rust shim a maintainer would add instead).  Names and
// behaviour follow the Rust type: struct fields spam_csr/src/lib.rs:25-32, Matrix trait
// spam_matrix/src/lib.rs:15-27, identity lib.rs:177-185, mul_hash mul_hash.rs:13-36, Mul lib.rs:292-297.
// Where the reference panics this throws spam::Panic.
#pragma once

#include <algorithm>
#include <complex>
#include <cstdint>
#include <map>
#include <memory>
#include <utility>
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

// Element types.  The device holds f32, f64 and i64 (= Wrapping<i64>).  The reference's other instantiations reach them
// from here:
//   * int8_t / int16_t / int32_t stand for Wrapping<i8> (the type of the reference's own mul_hash test, tests.rs:358),
//     Wrapping<i16>, Wrapping<i32>: Z/2^64 -> Z/2^k is a ring homomorphism, so the values run as i64 with 64-bit
//     wrapping on the device and keep their low bits on the way back — exact;
//   * std::complex<float / double> (the fuzz targets and MatrixType::Complex, lib.rs:428-430): products and sums are
//     bilinear, C = (Ar Br - Ai Bi) + i (Ar Bi + Ai Br) from four real products over ONE pattern (mul_hash's pattern
//     depends on the indices only and keeps computed zeros; the four are taken with sorted rows so that their value
//     arrays line up).  Same result as complex arithmetic up to rounding, at four times the real cost.
namespace detail {
template <class T> struct is_complex : std::false_type {};
template <class R> struct is_complex<std::complex<R>> : std::true_type { typedef R real; };
template <class T> struct device_type { typedef T type; };
template <> struct device_type<int8_t> { typedef int64_t type; };
template <> struct device_type<int16_t> { typedef int64_t type; };
template <> struct device_type<int32_t> { typedef int64_t type; };
template <class D, class T>
std::vector<D> widen(const std::vector<T>& v) { return std::vector<D>(v.begin(), v.end()); }  // (sign-extends)
template <class T, class D>
std::vector<T> narrow(const std::vector<D>& v) {
  std::vector<T> o(v.size());
  for (size_t i = 0; i < v.size(); ++i) o[i] = (T)(typename std::make_unsigned<D>::type)v[i];  // low bits: Wrapping<T>
  return o;
}
// the typed entry points of one device type
template <class D> struct Abi;
template <> struct Abi<double> {
  static int upload(uint64_t r, uint64_t c, uint64_t n, const uint64_t* o, const uint64_t* i, const double* v, int s, spam_csr** h) { return spam_csr_upload_f64(r, c, n, o, i, v, s, h); }
  static int numeric(spam_spgemm_plan* p, int sort, uint64_t* o, uint64_t* i, double* v) { return spam_spgemm_numeric_f64(p, sort, o, i, v); }
  static int spmv(const spam_csr* a, const double* x, double* y) { return spam_spmv_f64(a, x, y); }
  static int download(const spam_csr* a, uint64_t* o, uint64_t* i, double* v) { return spam_csr_download_f64(a, o, i, v); }
};
template <> struct Abi<float> {
  static int upload(uint64_t r, uint64_t c, uint64_t n, const uint64_t* o, const uint64_t* i, const float* v, int s, spam_csr** h) { return spam_csr_upload_f32(r, c, n, o, i, v, s, h); }
  static int numeric(spam_spgemm_plan* p, int sort, uint64_t* o, uint64_t* i, float* v) { return spam_spgemm_numeric_f32(p, sort, o, i, v); }
  static int spmv(const spam_csr* a, const float* x, float* y) { return spam_spmv_f32(a, x, y); }
  static int download(const spam_csr* a, uint64_t* o, uint64_t* i, float* v) { return spam_csr_download_f32(a, o, i, v); }
};
template <> struct Abi<int64_t> {
  static int upload(uint64_t r, uint64_t c, uint64_t n, const uint64_t* o, const uint64_t* i, const int64_t* v, int s, spam_csr** h) { return spam_csr_upload_i64(r, c, n, o, i, v, s, h); }
  static int numeric(spam_spgemm_plan* p, int sort, uint64_t* o, uint64_t* i, int64_t* v) { return spam_spgemm_numeric_i64(p, sort, o, i, v); }
  static int spmv(const spam_csr* a, const int64_t* x, int64_t* y) { return spam_spmv_i64(a, x, y); }
  static int download(const spam_csr* a, uint64_t* o, uint64_t* i, int64_t* v) { return spam_csr_download_i64(a, o, i, v); }
};
}  // namespace detail

template <class T, bool IS_SORTED>
class CsrMatrix {
  static constexpr bool kComplex = detail::is_complex<T>::value;
  typedef typename detail::device_type<T>::type D;  // what the device holds (complex: see real_imag())
  static constexpr bool kWidened = !std::is_same<T, D>::value;
  static_assert(kComplex || std::is_same<D, float>::value || std::is_same<D, double>::value || std::is_same<D, int64_t>::value,
                "element types: f32, f64, i64, i8 / i16 / i32 (as i64), complex<f32 / f64> (as four real products)");

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
    part_[0] = std::move(o.part_[0]), part_[1] = std::move(o.part_[1]);
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

  // DokMatrix's entries: BTreeMap<(row, col), T> (spam_dok/src/lib.rs:19-23); zeros are never stored (:171-175)
  typedef std::map<std::pair<uint64_t, uint64_t>, T> Dok;

  // impl From<DokMatrix<T>> for CsrMatrix<T, true>  (lib.rs:315-334): rows in column order
  static CsrMatrix from_dok(uint64_t rows, uint64_t cols, const Dok& entries) {
    static_assert(IS_SORTED, "From<DokMatrix> gives sorted rows; the shuffled constructor takes an rng");
    return build_from_dok(rows, cols, entries, [](std::vector<std::pair<uint64_t, T>>&) {});
  }
  // CsrMatrix::<T, false>::from_dok(dok, rng)  (lib.rs:336-358): every row's entries in a random order
  template <class Rng>
  static CsrMatrix from_dok(uint64_t rows, uint64_t cols, const Dok& entries, Rng& rng) {
    static_assert(!IS_SORTED, "from_dok(dok, rng) builds CsrMatrix<T, false>");
    return build_from_dok(rows, cols, entries, [&](std::vector<std::pair<uint64_t, T>>& row) {
      for (size_t i = row.size(); i > 1; --i) std::swap(row[i - 1], row[(size_t)(rng() % i)]);
    });
  }
  // impl From<CsrMatrix> for DokMatrix  (lib.rs:375-384) through set_element: zeros dropped
  Dok to_dok() const {
    Dok out;
    for (uint64_t r = 0; r < rows_; ++r)
      for (uint64_t p = offsets_[r]; p < offsets_[r + 1]; ++p) {
        if (vals_[p] != T(0)) out[{r, indices_[p]}] = vals_[p];
        else out.erase({r, indices_[p]});
      }
    return out;
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
    if constexpr (kComplex) {
      return part(false).invariants();  // (same structure)
    } else {
      int ok = 0;
      try {
        check(spam_csr_invariants(device(), &ok));
      } catch (const Panic& p) {
        if (p.code == SPAM_E_INVALID || p.code == SPAM_E_INDEX_RANGE) return false;
        throw;
      }
      return ok != 0;
    }
  }

  // pub fn mul_hash<const B1: bool, const B2: bool>(&self, rhs: &CsrMatrix<T, B1>) -> CsrMatrix<T, B2>
  template <bool B1, bool B2>
  CsrMatrix<T, B2> mul_hash(const CsrMatrix<T, B1>& rhs) const {
    if constexpr (kComplex) {
      // four real products with sorted rows (a sorted row is one of the orders B2 = false allows)
      const auto &ar = part(false), &ai = part(true);
      const auto &br = rhs.part(false), &bi = rhs.part(true);
      auto rr = ar.template mul_hash<B1, true>(br), ii = ai.template mul_hash<B1, true>(bi);
      auto ri = ar.template mul_hash<B1, true>(bi), ir = ai.template mul_hash<B1, true>(br);
      if (rr.indices() != ii.indices() || rr.indices() != ri.indices() || rr.indices() != ir.indices() ||
          rr.offsets() != ii.offsets() || rr.offsets() != ri.offsets() || rr.offsets() != ir.offsets())
        throw Panic(SPAM_E_INVALID, "the real products of a complex product came back with different patterns");
      std::vector<T> vals(rr.nnz());
      for (size_t k = 0; k < vals.size(); ++k) vals[k] = T(rr.vals()[k] - ii.vals()[k], ri.vals()[k] + ir.vals()[k]);
      return CsrMatrix<T, B2>(rows_, rhs.cols(), std::move(vals), rr.indices(), rr.offsets());
    } else {
      spam_spgemm_plan* plan = nullptr;
      uint64_t nnz = 0;
      check(spam_spgemm_symbolic(device(), rhs.device(), &plan, &nnz));
      std::vector<uint64_t> offsets(rows_ + 1), indices(nnz);  // Vec::with_capacity(nnz) (mul_hash.rs:119)
      std::vector<D> vals(nnz);
      const int st = detail::Abi<D>::numeric(plan, B2 ? 1 : 0, offsets.data(), indices.data(), vals.data());
      spam_spgemm_plan_free(plan);
      check(st);
      return CsrMatrix<T, B2>(rows_, rhs.cols(), from_device(std::move(vals)), std::move(indices), std::move(offsets));
    }
  }

  // impl Mul for &CsrMatrix<T, B>  ->  CsrMatrix<T, false>  (lib.rs:292-297)
  CsrMatrix<T, false> operator*(const CsrMatrix& rhs) const { return mul_hash<IS_SORTED, false>(rhs); }

  // impl Add / Sub for CsrMatrix (lib.rs:276-290) through apply_elementwise (lib.rs:83-149): sorted-union rows,
  // results equal to zero are kept
  CsrMatrix<T, true> operator+(const CsrMatrix& rhs) const { return ewise(rhs, 0); }
  CsrMatrix<T, true> operator-(const CsrMatrix& rhs) const { return ewise(rhs, 1); }

  // the SpMV analysis on the uploaded copy (spam_spmv_prepare): worth it when the matrix is multiplied many times
  void spmv_prepare() const {
    if constexpr (kComplex) {  // a complex matrix multiplies through its two real parts: prepare both (they are cached)
      part(false).spmv_prepare();
      part(true).spmv_prepare();
    } else {
      check(spam_spmv_prepare(const_cast<spam_csr*>(device()), nullptr));
    }
  }

  // y = A x (absent upstream; mul_hash with an n x 1 right-hand side)
  std::vector<T> spmv(const std::vector<T>& x) const {
    if (x.size() != cols_) throw Panic(SPAM_E_INVALID, "x.len() != cols");
    if constexpr (kComplex) {  // y = (Ar xr - Ai xi) + i (Ar xi + Ai xr)
      typedef typename detail::is_complex<T>::real R;
      std::vector<R> xr(x.size()), xi(x.size());
      for (size_t k = 0; k < x.size(); ++k) xr[k] = x[k].real(), xi[k] = x[k].imag();
      const auto &ar = part(false), &ai = part(true);
      const auto rr = ar.spmv(xr), ii = ai.spmv(xi), ri = ar.spmv(xi), ir = ai.spmv(xr);
      std::vector<T> y(rows_);
      for (size_t k = 0; k < y.size(); ++k) y[k] = T(rr[k] - ii[k], ri[k] + ir[k]);
      return y;
    } else {
      std::vector<D> y(rows_);
      if constexpr (kWidened) {
        const std::vector<D> xd = detail::widen<D>(x);
        check(detail::Abi<D>::spmv(device(), xd.data(), y.data()));
      } else {
        check(detail::Abi<D>::spmv(device(), x.data(), y.data()));
      }
      return from_device(std::move(y));
    }
  }

  // the uploaded copy (created on first use, owned by this object)
  const spam_csr* device() const {
    static_assert(!kComplex, "a complex matrix holds no device handle of its own: see part()");
    if (!dev_) {
      if constexpr (kWidened) {
        const std::vector<D> wide = detail::widen<D>(vals_);
        check(detail::Abi<D>::upload(rows_, cols_, indices_.size(), offsets_.data(), indices_.data(), wide.data(), IS_SORTED, &dev_));
      } else {
        check(detail::Abi<D>::upload(rows_, cols_, indices_.size(), offsets_.data(), indices_.data(), vals_.data(), IS_SORTED, &dev_));
      }
    }
    return dev_;
  }

  // complex values: the real (imag = false) or imaginary part as a real matrix over the same pattern.  Built on first
  // use and kept, like dev_: its device handle and SpMV plan then survive from one product to the next (every complex
  // mul_hash / spmv / + used to upload two to four real matrices again).
  const auto& part(bool imag) const {
    static_assert(kComplex, "part() is for complex matrices");
    typedef typename detail::is_complex<T>::real R;
    typedef CsrMatrix<R, IS_SORTED> P;
    std::shared_ptr<void>& slot = part_[imag ? 1 : 0];
    if (!slot) {
      std::vector<R> v(vals_.size());
      for (size_t k = 0; k < v.size(); ++k) v[k] = imag ? vals_[k].imag() : vals_[k].real();
      slot = std::make_shared<P>(rows_, cols_, std::move(v), indices_, offsets_);
    }
    return *static_cast<const P*>(slot.get());
  }

 private:
  template <class RowOrder>
  static CsrMatrix build_from_dok(uint64_t rows, uint64_t cols, const Dok& entries, RowOrder&& order) {
    std::vector<T> vals;
    std::vector<uint64_t> idx, off(rows + 1, 0);
    std::vector<std::pair<uint64_t, T>> row;
    auto it = entries.begin();
    for (uint64_t r = 0; r < rows; ++r) {
      row.clear();
      for (; it != entries.end() && it->first.first == r; ++it) {
        if (it->first.second >= cols) throw Panic(SPAM_E_INVALID, "IndexError");  // spam_matrix::IndexError
        if (it->second != T(0)) row.emplace_back(it->first.second, it->second);
      }
      order(row);
      for (const auto& e : row) idx.push_back(e.first), vals.push_back(e.second);
      off[r + 1] = idx.size();
    }
    if (it != entries.end()) throw Panic(SPAM_E_INVALID, "IndexError");
    return CsrMatrix(rows, cols, std::move(vals), std::move(idx), std::move(off));
  }
  // device values in this matrix's element type (i64 -> i8 / i16 / i32 keeps the low bits)
  static std::vector<T> from_device(std::vector<D>&& v) {
    if constexpr (kWidened) return detail::narrow<T>(v);
    else return std::move(v);
  }
  CsrMatrix<T, true> ewise(const CsrMatrix& rhs, int subtract) const {
    if constexpr (kComplex) {  // (a + ib) +- (c + id), part by part; zeros are kept, so one pattern
      auto re = subtract ? part(false) - rhs.part(false) : part(false) + rhs.part(false);
      auto im = subtract ? part(true) - rhs.part(true) : part(true) + rhs.part(true);
      if (re.indices() != im.indices() || re.offsets() != im.offsets())
        throw Panic(SPAM_E_INVALID, "the real sums of a complex sum came back with different patterns");
      std::vector<T> vals(re.nnz());
      for (size_t k = 0; k < vals.size(); ++k) vals[k] = T(re.vals()[k], im.vals()[k]);
      return CsrMatrix<T, true>(rows_, cols_, std::move(vals), re.indices(), re.offsets());
    } else {
      spam_csr* c = nullptr;
      check(spam_csr_add(device(), rhs.device(), subtract, &c));
      uint64_t rows = 0, cols = 0, nnz = 0;
      check(spam_csr_dims(c, &rows, &cols, &nnz, nullptr, nullptr));
      std::vector<uint64_t> offsets(rows + 1), indices(nnz);
      std::vector<D> vals(nnz);
      const int st = detail::Abi<D>::download(c, offsets.data(), indices.data(), vals.data());
      spam_csr_free(c);
      check(st);
      return CsrMatrix<T, true>(rows, cols, from_device(std::move(vals)), std::move(indices), std::move(offsets));
    }
  }
  void release() {
    if (dev_) spam_csr_free(dev_);
    dev_ = nullptr;
  }
  uint64_t rows_ = 0, cols_ = 0;
  std::vector<T> vals_;
  std::vector<uint64_t> indices_, offsets_;
  mutable spam_csr* dev_ = nullptr;
  mutable std::shared_ptr<void> part_[2];  // complex matrices: the cached real / imaginary parts (CsrMatrix<R, IS_SORTED>)
};

}  // namespace spam
