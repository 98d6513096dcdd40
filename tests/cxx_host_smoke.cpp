// C++ host-side smoke test of include/spam_csr.hpp against libspam_hip.so (run by tests/test_cxx_host_gpu.py).
// Mirrors the reference's identity / mul_hash usage: I*A == A, (A*A) has the banded pattern, spmv(A, 1) = row sums.
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include "spam_csr.hpp"

// The reference's other element types through the C++ mirror: Wrapping<i8> (tests.rs:358) runs as i64 and keeps its low
// byte — compared EXACTLY with the triple loop (spam_dok/src/lib.rs:206-233) in 8-bit wrapping arithmetic; Complex<f64>
// runs as four real products — compared with the triple loop in complex arithmetic.
namespace {
struct Lcg {
  uint64_t s;
  uint32_t next() { return (uint32_t)((s = s * 6364136223846793005ull + 1442695040888963407ull) >> 33); }
};
template <class T, class Draw>
spam::CsrMatrix<T, true> random_csr(uint64_t rows, uint64_t cols, unsigned fill_pct, Lcg& g, Draw draw, std::vector<T>* dense,
                                    std::vector<char>* present) {
  std::vector<T> vals;
  std::vector<uint64_t> idx, off{0};
  dense->assign(rows * cols, T(0)), present->assign(rows * cols, 0);
  for (uint64_t i = 0; i < rows; ++i) {
    for (uint64_t j = 0; j < cols; ++j)
      if (g.next() % 100 < fill_pct) {
        const T v = draw(g);
        idx.push_back(j), vals.push_back(v);
        (*dense)[i * cols + j] = v, (*present)[i * cols + j] = 1;
      }
    off.push_back(idx.size());
  }
  return spam::CsrMatrix<T, true>(rows, cols, vals, idx, off);
}
int8_t wrap8(int v) { return (int8_t)(uint8_t)(unsigned)v; }

// tests.rs:356-371 as written: unsorted CSR operands from DOK maps, mul_hash::<false, false>, the product compared as a
// zero-dropped DOK map with the DOK triple loop in Wrapping<i8> arithmetic.
int reference_property_wrapping_i8() {
  typedef spam::CsrMatrix<int8_t, false> M;
  Lcg g{777};
  auto rng = [&]() { return (uint64_t)g.next(); };
  for (int rep = 0; rep < 60; ++rep) {
    const uint64_t hi = rep % 3 == 2 ? 24 : 4;
    const uint64_t l = 1 + g.next() % hi, m = 1 + g.next() % hi, n = 1 + g.next() % hi;
    auto rand_dok = [&](uint64_t r, uint64_t c) {
      M::Dok d;
      for (uint64_t k = g.next() % (2 * r * c + 1); k > 0; --k) {
        const int8_t v = (int8_t)(uint8_t)(g.next() & 255u);
        if (v != 0) d[{g.next() % r, g.next() % c}] = v;  // (DokMatrix never stores a zero)
      }
      return d;
    };
    const M::Dok da = rand_dok(l, m), db = rand_dok(m, n);
    const M A = M::from_dok(l, m, da, rng), B = M::from_dok(m, n, db, rng);
    const M C = A.mul_hash<false, false>(B);
    if (!C.invariants()) return std::puts("P2: invariants(C) failed"), 1;
    M::Dok want;  // t = t + a[i,k] * b[k,j], k ascending (spam_dok/src/lib.rs:214-229); zeros are not stored
    for (uint64_t i = 0; i < l; ++i)
      for (uint64_t j = 0; j < n; ++j) {
        int8_t t = 0;
        for (uint64_t k = 0; k < m; ++k) {
          const auto a = da.find({i, k});
          const auto b = db.find({k, j});
          if (a != da.end() && b != db.end()) t = wrap8((int)t + (int)wrap8((int)a->second * (int)b->second));
        }
        if (t != 0) want[{i, j}] = t;
      }
    if (C.to_dok() != want) return std::printf("P2 (Wrapping<i8>): case %d differs\n", rep), 1;
  }
  return 0;
}

// tests.rs `new` (:127-143), `identity` (:145-160), `from_arb_dok` (:179-197) and `add` (:334-354), for both IS_SORTED:
// every CSR result has its invariants and, converted back to DOK, equals what the DOK side computes (test_commutes).
int reference_base_cases_and_add() {
  typedef spam::CsrMatrix<int8_t, false> MU;
  typedef spam::CsrMatrix<int8_t, true> MS;
  Lcg g{4242};
  auto rng = [&]() { return (uint64_t)g.next(); };
  for (int rep = 0; rep < 40; ++rep) {
    const uint64_t m = 1 + g.next() % 24, n = 1 + g.next() % 24;
    const MU nu = MU::new_(m, n);
    const MS ns = MS::new_(m, n);
    if (!nu.invariants() || !ns.invariants() || !nu.to_dok().empty() || !ns.to_dok().empty()) return std::puts("new: not the empty DOK"), 1;
    MS::Dok eye;
    for (uint64_t i = 0; i < m; ++i) eye[{i, i}] = 1;
    const MU iu = MU::identity(m);
    const MS is = MS::identity(m);
    if (!iu.invariants() || !is.invariants() || iu.to_dok() != eye || is.to_dok() != eye) return std::puts("identity: not DokMatrix::identity"), 1;
    auto rand_dok = [&]() {
      MS::Dok d;
      for (uint64_t k = g.next() % (2 * m * n + 1); k > 0; --k) {
        const int8_t v = (int8_t)(uint8_t)(g.next() & 255u);
        if (v != 0) d[{g.next() % m, g.next() % n}] = v;
      }
      return d;
    };
    const MS::Dok d1 = rand_dok(), d2 = rand_dok();
    const MU u1 = MU::from_dok(m, n, d1, rng), u2 = MU::from_dok(m, n, d2, rng);
    const MS s1 = MS::from_dok(m, n, d1), s2 = MS::from_dok(m, n, d2);
    if (!u1.invariants() || !s1.invariants() || u1.to_dok() != d1 || s1.to_dok() != d1) return std::puts("from_dok: round trip differs"), 1;
    MS::Dok sum = d1;  // DokMatrix + DokMatrix in Wrapping<i8>; a zero result is not stored
    for (const auto& e : d2) {
      const int8_t t = wrap8((int)(sum.count(e.first) ? sum[e.first] : (int8_t)0) + (int)e.second);
      if (t != 0) sum[e.first] = t;
      else sum.erase(e.first);
    }
    const auto cu = u1 + u2;
    const auto cs = s1 + s2;
    if (!cu.invariants() || !cs.invariants() || cu.to_dok() != sum || cs.to_dok() != sum) return std::printf("add: case %d differs\n", rep), 1;
  }
  return 0;
}

int other_element_types() {
  Lcg g{12345};
  for (int rep = 0; rep < 20; ++rep) {
    const uint64_t l = 1 + g.next() % 24, m = 1 + g.next() % 24, n = 1 + g.next() % 24;
    // ---- Wrapping<i8>
    std::vector<int8_t> ad, bd;
    std::vector<char> ap, bp;
    auto draw8 = [](Lcg& r) { return (int8_t)(uint8_t)(r.next() & 255u); };  // zeros included: stored entries may be 0
    auto A = random_csr<int8_t>(l, m, 40, g, draw8, &ad, &ap);
    auto B = random_csr<int8_t>(m, n, 40, g, draw8, &bd, &bp);
    auto C = A.mul_hash<true, true>(B);
    if (!C.invariants()) return std::puts("i8: invariants(C) failed"), 1;
    uint64_t p = 0;
    for (uint64_t i = 0; i < l; ++i) {
      for (uint64_t j = 0; j < n; ++j) {
        int8_t t = 0;
        bool any = false;
        for (uint64_t k = 0; k < m; ++k)
          if (ap[i * m + k] && bp[k * n + j]) any = true, t = wrap8((int)t + (int)wrap8((int)ad[i * m + k] * (int)bd[k * n + j]));
        if (!any) continue;  // mul_hash's pattern: the symbolic one, computed zeros kept
        if (p >= C.nnz() || C.indices()[p] != j || C.vals()[p] != t) return std::printf("i8: C(%llu,%llu) wrong\n", (unsigned long long)i, (unsigned long long)j), 1;
        ++p;
      }
      if (C.offsets()[i + 1] != p) return std::puts("i8: row length wrong"), 1;
    }
    std::vector<int8_t> x(m);
    for (auto& v : x) v = draw8(g);
    const auto y = A.spmv(x);
    for (uint64_t i = 0; i < l; ++i) {
      int8_t t = 0;
      for (uint64_t k = 0; k < m; ++k) t = wrap8((int)t + (int)wrap8((int)ad[i * m + k] * (int)x[k]));
      if (y[i] != t) return std::puts("i8: spmv wrong"), 1;
    }
    const auto S = A + A;
    if (S.indices() != A.indices()) return std::puts("i8: A + A pattern"), 1;
    for (uint64_t q = 0; q < S.nnz(); ++q)
      if (S.vals()[q] != wrap8(2 * (int)A.vals()[q])) return std::puts("i8: A + A values"), 1;
    // ---- Complex<f64>
    typedef std::complex<double> Z;
    std::vector<Z> zad, zbd;
    auto drawz = [](Lcg& r) { return Z((double)(r.next() % 2001) / 1000.0 - 1.0, (double)(r.next() % 2001) / 1000.0 - 1.0); };
    auto ZA = random_csr<Z>(l, m, 40, g, drawz, &zad, &ap);
    auto ZB = random_csr<Z>(m, n, 40, g, drawz, &zbd, &bp);
    auto ZC = ZA * ZB;  // CsrMatrix<Z, false>; its rows come out sorted
    if (!ZC.invariants()) return std::puts("complex: invariants(C) failed"), 1;
    p = 0;
    for (uint64_t i = 0; i < l; ++i)
      for (uint64_t j = 0; j < n; ++j) {
        Z t(0, 0);
        double scale = 0;
        bool any = false;
        for (uint64_t k = 0; k < m; ++k)
          if (ap[i * m + k] && bp[k * n + j]) any = true, t += zad[i * m + k] * zbd[k * n + j], scale += std::abs(zad[i * m + k]) * std::abs(zbd[k * n + j]);
        if (!any) continue;
        if (p >= ZC.nnz() || ZC.indices()[p] != j || std::abs(ZC.vals()[p] - t) > 1e-13 * scale) return std::puts("complex: C entry wrong"), 1;
        ++p;
      }
    if (p != ZC.nnz()) return std::puts("complex: nnz(C) wrong"), 1;
    std::vector<Z> zx(m);
    for (auto& v : zx) v = drawz(g);
    const auto zy = ZA.spmv(zx);
    for (uint64_t i = 0; i < l; ++i) {
      Z t(0, 0);
      for (uint64_t k = 0; k < m; ++k) t += zad[i * m + k] * zx[k];
      if (std::abs(zy[i] - t) > 1e-12) return std::puts("complex: spmv wrong"), 1;
    }
    const auto ZD = ZA - ZA;
    if (ZD.indices() != ZA.indices()) return std::puts("complex: A - A lost structure"), 1;
    for (const Z& v : ZD.vals())
      if (v != Z(0, 0)) return std::puts("complex: A - A != 0"), 1;
  }
  return 0;
}
}  // namespace

int main() {
  try {
    spam::check(spam_hip_init(0));
    const uint64_t n = 1000;
    std::vector<double> vals;
    std::vector<uint64_t> idx, off{0};
    for (uint64_t i = 0; i < n; ++i) {  // tridiagonal
      for (int d = -1; d <= 1; ++d) {
        long j = (long)i + d;
        if (j < 0 || j >= (long)n) continue;
        idx.push_back((uint64_t)j);
        vals.push_back(1.0 + 0.001 * (double)i + 0.5 * d);
      }
      off.push_back(idx.size());
    }
    spam::CsrMatrix<double, true> A(n, n, vals, idx, off);
    if (!A.invariants()) return std::puts("invariants(A) failed"), 1;
    auto I = spam::CsrMatrix<double, true>::identity(n);
    auto IA = I.mul_hash<true, true>(A);
    if (IA.offsets() != A.offsets() || IA.indices() != A.indices() || IA.vals() != A.vals()) return std::puts("I*A != A"), 1;
    auto C = A * A;  // CsrMatrix<double, false>
    if (C.nnz() != 5 * n - 6 || !C.invariants()) return std::puts("A*A pattern wrong"), 1;
    auto y = A.spmv(std::vector<double>(n, 1.0));
    for (uint64_t i = 0; i < n; ++i) {
      double s = 0;
      for (uint64_t p = off[i]; p < off[i + 1]; ++p) s = (p == off[i]) ? vals[p] : s + vals[p];
      if (y[i] != s) return std::printf("spmv row %llu: %g != %g\n", (unsigned long long)i, y[i], s), 1;
    }
    auto Z = A - A;  // every entry cancels: structure kept (explicit zeros), as apply_elementwise does
    if (Z.indices() != A.indices() || Z.nnz() != A.nnz()) return std::puts("A - A lost structure"), 1;
    for (double v : Z.vals())
      if (v != 0.0) return std::puts("A - A != 0"), 1;
    auto S = A + I;
    if (S.nnz() != A.nnz() || !S.invariants()) return std::puts("A + I pattern wrong"), 1;
    bool threw = false;
    try {
      auto B = spam::CsrMatrix<double, true>::identity(n + 1);
      (void)A.mul_hash<true, false>(B);
    } catch (const spam::Panic& p) {
      threw = p.code == SPAM_E_INVALID;
    }
    if (!threw) return std::puts("dimension mismatch did not panic"), 1;
    if (int rc = other_element_types()) return rc;
    if (int rc = reference_property_wrapping_i8()) return rc;
    if (int rc = reference_base_cases_and_add()) return rc;
    std::puts("cxx host smoke ok");
    return 0;
  } catch (const spam::Panic& p) {
    std::printf("panic %d: %s\n", p.code, p.what());
    return 2;
  }
}
