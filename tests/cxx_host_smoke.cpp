// C++ host-side smoke test of include/spam_csr.hpp against libspam_hip.so (run by tests/test_cxx_host_gpu.py).
// Mirrors the reference's identity / mul_hash usage: I*A == A, (A*A) has the banded pattern, spmv(A, 1) = row sums.
#include <cstdio>
#include <vector>

#include "spam_csr.hpp"

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
    std::puts("cxx host smoke ok");
    return 0;
  } catch (const spam::Panic& p) {
    std::printf("panic %d: %s\n", p.code, p.what());
    return 2;
  }
}
