// matrix.hpp — host-side dense helpers with the reference's call signatures.
//
// Drop-in for the free functions of the reference's include/matrix.hpp (mov :10/:18, add :26/:34, sub :42/:50,
// mul :58/:66/:74/:95, div :122/:131, norm :140, dot :151, sign :162, linsolve :166).  The example programs
// call mul(double*, const double*, double, int16_t) and add(double*, const double*, const double*, int16_t)
// on the plant state (<example>/main.cpp:72-73), so the names, argument order and rounding behaviour are kept:
//   * div multiplies by the reciprocal (one rounding for 1/c, then one per element);
//   * norm / dot accumulate in index order;  sign(0) = +1;
//   * linsolve is Gaussian elimination with partial pivoting on a COLUMN-major matrix, overwrites both arguments.
// None of this is on the accelerated path: the controller tick runs inside libcgmres_hip.so.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace cgmres_detail {

template <class Op>
inline void map1(double* out, const double* a, int32_t count, Op op) {
  for (int32_t k = 0; k < count; ++k) out[k] = op(a[k]);
}
template <class Op>
inline void map2(double* out, const double* a, const double* b, int32_t count, Op op) {
  for (int32_t k = 0; k < count; ++k) out[k] = op(a[k], b[k]);
}
inline void no_alias(const void* out, const void* in, const char* who) {
  if (out == in) {
    fprintf(stderr, "%s: output aliases an input\n", who);
    exit(-1);
  }
}

}  // namespace cgmres_detail

// ret = src
inline void mov(double* ret, const double* vec, const int16_t row) {
  cgmres_detail::map1(ret, vec, row, [](double v) { return v; });
}
inline void mov(double* ret, const double* mat, const int16_t row, const int16_t col) {
  cgmres_detail::map1(ret, mat, int32_t(row) * col, [](double v) { return v; });
}

// ret = a + b
inline void add(double* ret, const double* vec1, const double* vec2, const int16_t row) {
  cgmres_detail::map2(ret, vec1, vec2, row, [](double p, double q) { return p + q; });
}
inline void add(double* ret, const double* mat1, const double* mat2, const int16_t row, const int16_t col) {
  cgmres_detail::map2(ret, mat1, mat2, int32_t(row) * col, [](double p, double q) { return p + q; });
}

// ret = a - b
inline void sub(double* ret, const double* vec1, const double* vec2, const int16_t row) {
  cgmres_detail::map2(ret, vec1, vec2, row, [](double p, double q) { return p - q; });
}
inline void sub(double* ret, const double* mat1, const double* mat2, const int16_t row, const int16_t col) {
  cgmres_detail::map2(ret, mat1, mat2, int32_t(row) * col, [](double p, double q) { return p - q; });
}

// ret = a * c
inline void mul(double* ret, const double* vec, const double c, const int16_t row) {
  cgmres_detail::map1(ret, vec, row, [c](double v) { return v * c; });
}
inline void mul(double* ret, const double* mat, const double c, const int16_t row, const int16_t col) {
  cgmres_detail::map1(ret, mat, int32_t(row) * col, [c](double v) { return v * c; });
}

// ret = mat * vec, mat column-major row x col; accumulation column by column starting from zero
inline void mul(double* ret, const double* mat, const double* vec, const int16_t row, const int16_t col) {
  cgmres_detail::no_alias(ret, vec, "mul(mat,vec)");
  for (int32_t r = 0; r < row; ++r) ret[r] = 0.0;
  for (int32_t c = 0; c < col; ++c) {
    const double* column = mat + int32_t(row) * c;
    const double w = vec[c];
    for (int32_t r = 0; r < row; ++r) ret[r] += column[r] * w;
  }
}

// ret (row x col) = mat1 (row x l) * mat2 (l x col); the reference indexes all three with the leading dimension
// `col` (matrix.hpp:104-118), kept as-is.
inline void mul(double* ret, const double* mat1, const double* mat2, const int16_t l, const int16_t row,
                const int16_t col) {
  cgmres_detail::no_alias(ret, mat1, "mul(mat,mat)");
  cgmres_detail::no_alias(ret, mat2, "mul(mat,mat)");
  for (int32_t r = 0; r < row; ++r) {
    double* out = ret + int32_t(col) * r;
    for (int32_t c = 0; c < col; ++c) out[c] = 0;
    for (int32_t k = 0; k < l; ++k) {
      const double w = mat1[int32_t(col) * r + k];
      const double* in = mat2 + int32_t(col) * k;
      for (int32_t c = 0; c < col; ++c) out[c] += w * in[c];
    }
  }
}

// ret = a / c, evaluated as a * (1/c)
inline void div(double* ret, const double* vec, const double c, const int16_t row) {
  const double r = 1.0 / c;
  cgmres_detail::map1(ret, vec, row, [r](double v) { return v * r; });
}
inline void div(double* ret, const double* mat, const double c, const int16_t row, const int16_t col) {
  const double r = 1.0 / c;
  cgmres_detail::map1(ret, mat, int32_t(row) * col, [r](double v) { return v * r; });
}

// Euclidean norm / inner product, sequential accumulation
inline double dot(const double* vec1, const double* vec2, const int16_t n) {
  double acc = 0;
  for (int32_t k = 0; k < n; ++k) acc += vec1[k] * vec2[k];
  return acc;
}
inline double norm(const double* vec, int16_t n) { return sqrt(dot(vec, vec, n)); }

inline double sign(const double x) { return (x < 0.0) ? -1.0 : 1.0; }

// vec <- mat \ vec ; mat is n x n column-major (entry (r,c) at mat[n*c + r]); both are destroyed
inline void linsolve(double* vec, double* mat, const int16_t n) {
  auto at = [mat, n](int32_t r, int32_t c) -> double& { return mat[int32_t(n) * c + r]; };
  for (int32_t k = 0; k + 1 < n; ++k) {
    int32_t piv = k;
    double best = fabs(at(k, k));
    for (int32_t r = k + 1; r < n; ++r) {
      const double cand = fabs(at(r, k));
      if (best < cand) best = cand, piv = r;
    }
    if (piv != k) {
      double t = vec[k];
      vec[k] = vec[piv];
      vec[piv] = t;
      for (int32_t c = k; c < n; ++c) {
        t = at(k, c);
        at(k, c) = at(piv, c);
        at(piv, c) = t;
      }
    }
    const double rp = 1.0 / at(k, k);
    for (int32_t r = k + 1; r < n; ++r) {
      at(r, k) = at(r, k) * rp;
      for (int32_t c = k + 1; c < n; ++c) at(r, c) -= at(r, k) * at(k, c);
      vec[r] -= at(r, k) * vec[k];
    }
  }
  for (int32_t r = n - 1; r >= 0; --r) {
    for (int32_t c = n - 1; c > r; --c) vec[r] -= at(r, c) * vec[c];
    vec[r] /= at(r, r);
  }
}
