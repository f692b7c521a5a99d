// colstats.hpp — column sums of a row-major [T x ncol] table in EXACTLY the order numpy uses for `values.std(axis=0)` of the reference's
// profile tables (voltage_control_env.py:70-72).
//
// `DataFrame.values` of a single-dtype frame is an F-ordered view, so numpy reduces every column along its CONTIGUOUS axis with
// pairwise summation (numpy/core/src/umath/loops.c.src, pairwise_sum: fewer than 8 elements sequentially; up to 128 with eight
// interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail; beyond that split at n/2 rounded
// down to a multiple of 8, recursively) — in chunks of 8192 elements whose pairwise sums are added one after the other (numpy_colsum).  At the real data's length (3 years of 3-minute rows = 526 080) a plain running sum differs
// from that by ~2e-12 relative — the noise scale of every env-step.  Here the same tree is walked over ROW ranges and every node holds
// one value per column, so the table is still read row-major (cache-friendly at 3 GB) while each column sees numpy's order.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace mapdn {

// out[c] = pairwise sum over rows [lo, lo + n) of f(tab[row * ncol + c], c)
template <class F>
inline void pairwise_colsum(const double* tab, int64_t lo, int64_t n, int ncol, F f, std::vector<double>& out) {
  out.assign((size_t)ncol, 0.0);
  if (n < 8) {
    for (int64_t i = 0; i < n; ++i) {
      const double* r = tab + (size_t)(lo + i) * ncol;
      for (int c = 0; c < ncol; ++c) out[(size_t)c] += f(r[c], c);
    }
    return;
  }
  if (n <= 128) {
    std::vector<double> acc((size_t)8 * ncol);
    for (int k = 0; k < 8; ++k) {
      const double* r = tab + (size_t)(lo + k) * ncol;
      for (int c = 0; c < ncol; ++c) acc[(size_t)k * ncol + c] = f(r[c], c);
    }
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) {
        const double* r = tab + (size_t)(lo + i + k) * ncol;
        double* a = &acc[(size_t)k * ncol];
        for (int c = 0; c < ncol; ++c) a[c] += f(r[c], c);
      }
    for (int c = 0; c < ncol; ++c) {
      const size_t s = (size_t)ncol, u = (size_t)c;
      out[u] = ((acc[u] + acc[s + u]) + (acc[2 * s + u] + acc[3 * s + u])) + ((acc[4 * s + u] + acc[5 * s + u]) + (acc[6 * s + u] + acc[7 * s + u]));
    }
    for (; i < n; ++i) {
      const double* r = tab + (size_t)(lo + i) * ncol;
      for (int c = 0; c < ncol; ++c) out[(size_t)c] += f(r[c], c);
    }
    return;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  std::vector<double> right;
  pairwise_colsum(tab, lo, n2, ncol, f, out);
  pairwise_colsum(tab, lo + n2, n - n2, ncol, f, right);
  for (int c = 0; c < ncol; ++c) out[(size_t)c] += right[(size_t)c];
}

// numpy's `add.reduce` along a contiguous axis: the ufunc machinery hands the inner loop at most NPY_BUFSIZE = 8192 elements at a time
// and adds each chunk's PAIRWISE sum to the running result, out = out + pairwise_sum(chunk) (measured on numpy 2.2: a single
// recursion over all T rows stops matching from T = 8193 on; chunks of 8192 match at every length tried, tests/test_colstats.py)
template <class F>
inline void numpy_colsum(const double* tab, int64_t T, int ncol, F f, std::vector<double>& out) {
  constexpr int64_t CHUNK = 8192;
  out.assign((size_t)ncol, 0.0);
  std::vector<double> part;
  for (int64_t lo = 0; lo < T; lo += CHUNK) {
    pairwise_colsum(tab, lo, T - lo < CHUNK ? T - lo : CHUNK, ncol, f, part);
    for (int c = 0; c < ncol; ++c) out[(size_t)c] += part[(size_t)c];
  }
}

// population std of every column / `div` (numpy's _var: mean = sum / T; x = a - mean; sum(x * x) / T; sqrt)
inline void column_std(const double* tab, int64_t T, int ncol, double div, std::vector<double>& stdv) {
  std::vector<double> mean, var;
  numpy_colsum(tab, T, ncol, [](double x, int) { return x; }, mean);
  for (int c = 0; c < ncol; ++c) mean[(size_t)c] /= (double)T;
  const double* m = mean.data();
  numpy_colsum(tab, T, ncol, [m](double x, int c) { const double d = x - m[c]; return d * d; }, var);
  stdv.resize((size_t)ncol);
  for (int c = 0; c < ncol; ++c) stdv[(size_t)c] = std::sqrt(var[(size_t)c] / (double)T) / div;
}

}  // namespace mapdn
