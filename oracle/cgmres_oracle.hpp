// ORACLE — TEST INFRASTRUCTURE ONLY. Never included by the product (cgmres_cpp_amd/, include/).
//
// CPU restatement of one C/GMRES control tick, operation for operation:
//   Cgmres<Model>::control   /root/reference/include/cgmres.hpp:78-110
//   Cgmres<Model>::F_func    /root/reference/include/cgmres.hpp:113-162
//   Cgmres<Model>::Ax_func   /root/reference/include/cgmres.hpp:164-175
//   Gmres::gmres             /root/reference/include/gmres.hpp:28-112
//   vector helpers           /root/reference/include/matrix.hpp (mov/add/sub/mul/div/norm/dot/sign/linsolve)
// Parity status: PINNED — tests/test_oracle_vs_ref.py compares this file bit-for-bit (fp64) with
// the unmodified reference compiled into oracle/_ref/, and tests/test_oracle_golden.py compares it
// with the committed fixtures tests/golden/*.npz that oracle/gen_golden.py produced from oracle/_ref.
//
// Differences from the reference that are deliberate and documented:
//   * dUdt(0) = 0 (the reference reads an uninitialised heap array on the first tick);
//   * dv / k_max / tol are run-time values;
//   * the executed Arnoldi count and exit reason are recorded (the reference drops them).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

#include "models.hpp"

namespace oracle {

enum ExitReason : int {
  kExitNatural = 0,     // ran all k_max Arnoldi iterations           gmres.hpp:46
  kExitConverged = 1,   // |rho_e[k+1]| < tol inside the loop          gmres.hpp:93-95
  kExitSmallResidual = 2,  // ||r0|| < tol, solution untouched         gmres.hpp:39-41
  kExitBreakdown = 3    // |h_{k+1,k}| < DBL_EPSILON, solution untouched  gmres.hpp:63-65
};

template <class Model, class T>
class Controller {
 public:
  static constexpr int nx = Model::dim_x, nu = Model::dim_u, np = Model::dim_p;

  Controller(int dv, int kmax, double tol)
      : dv_(dv), kmax_(kmax), len_(nu * dv), tol_(T(tol)), tun_(Model::tuning()),
        U_(len_, T(0)), dUdt_(len_, T(0)), Fh_(len_, T(0)), xh_(nx, T(0)), ptau_(np * (dv + 1) + 1, T(0)),
        V_(size_t(len_) * (kmax + 1), T(0)), H_(size_t(kmax + 1) * (kmax + 1), T(0)), rho_(kmax + 1, T(0)),
        g_(3 * kmax, T(0)), scratch_(len_, T(0)), xtau_(nx * (dv + 1)), ltau_(nx * (dv + 1)), b_(len_, T(0)),
        ub_(len_, T(0)) {}

  int dv() const { return dv_; }
  int kmax() const { return kmax_; }
  int len() const { return len_; }

  // cgmres.hpp:32-34
  T dtau_of(T t) const { return T(tun_.Tf) * (1 - std::exp(-T(tun_.alpha) * t)) / T(dv_); }

  void set_ptau(const T* p) {  // cgmres.hpp:36-39
    for (int i = 0; i < np * (dv_ + 1); ++i) ptau_[i] = p[i];
  }
  void set_ptau_repeat(const T* p) {  // cgmres.hpp:41-49
    for (int s = 0; s <= dv_; ++s)
      for (int j = 0; j < np; ++j) ptau_[np * s + j] = p[j];
  }
  void init_u0(const T* u0) {  // cgmres.hpp:51-59
    for (int s = 0; s < dv_; ++s)
      for (int j = 0; j < nu; ++j) U_[nu * s + j] = u0[j];
  }
  // cgmres.hpp:61-76 ; u0 is updated in place exactly like the reference does
  void init_u0_newton(T* u0, const T* x0, const T* p0, int n_loop) {
    T lmd0[nx], rhs[nu], mat[nu * nu];
    Model::dPhidx(lmd0, x0, p0);
    for (int it = 0; it < n_loop; ++it) {
      Model::dHdu(rhs, x0, u0, p0, lmd0);
      Model::ddHduu(mat, x0, u0, p0, lmd0);
      solve_in_place(rhs, mat, nu);
      for (int j = 0; j < nu; ++j) u0[j] = u0[j] - rhs[j];
    }
    init_u0(u0);
  }

  // One control tick — cgmres.hpp:78-110
  void control(T* u, const T* x) {
    const T h = T(tun_.h), zeta = T(tun_.zeta), dt = T(tun_.dt);
    std::vector<T>& b = b_;  // the reference's stack array b_vec (cgmres.hpp:79)
    // :83-85  x_dxh = dxdt*h + x  (scale, then add: two roundings)
    Model::dxdt(xh_.data(), x, &U_[0], &ptau_[0]);
    for (int i = 0; i < nx; ++i) xh_[i] = xh_[i] * h;
    for (int i = 0; i < nx; ++i) xh_[i] = xh_[i] + x[i];
    F(Fh_.data(), U_.data(), xh_.data(), t_ + h);  // :88
    F(b.data(), U_.data(), x, t_);                 // :91
    // :94-96  three separate passes; the division is a multiply by the reciprocal (matrix.hpp:122-128)
    const T c = (1 - zeta * h);
    for (int i = 0; i < len_; ++i) b[i] = b[i] * c;
    for (int i = 0; i < len_; ++i) b[i] = b[i] - Fh_[i];
    const T inv_h = T(1.0) / h;
    for (int i = 0; i < len_; ++i) b[i] = b[i] * inv_h;
    gmres(dUdt_.data(), b.data());  // :99
    // :102-103  U += dUdt*dt via a temporary
    for (int i = 0; i < len_; ++i) scratch_[i] = dUdt_[i] * dt;
    for (int i = 0; i < len_; ++i) U_[i] = U_[i] + scratch_[i];
    t_ = t_ + dt;  // :107
    for (int j = 0; j < nu; ++j) u[j] = U_[j];  // :109
  }

  // Optimality residual — cgmres.hpp:113-162
  void F(T* ret, const T* U, const T* x, T t) {
    const T dtau = dtau_of(t);  // :127
    T* X = xtau_.data();
    T* Lm = ltau_.data();
    for (int i = 0; i < nx; ++i) X[i] = x[i];  // :132
    for (int s = 0; s < dv_; ++s) {            // :133-140
      T* nxt = &X[nx * (s + 1)];
      Model::dxdt(nxt, &X[nx * s], &U[nu * s], &ptau_[np * s]);
      for (int i = 0; i < nx; ++i) nxt[i] = nxt[i] * dtau;
      for (int i = 0; i < nx; ++i) nxt[i] = nxt[i] + X[nx * s + i];
    }
    Model::dPhidx(&Lm[nx * dv_], &X[nx * dv_], &ptau_[np * dv_]);  // :145
    for (int s = dv_ - 1; s >= 0; --s) {                             // :146-153
      T* cur = &Lm[nx * s];
      Model::dHdx(cur, &X[nx * s], &U[nu * s], &ptau_[np * s], &Lm[nx * (s + 1)]);
      for (int i = 0; i < nx; ++i) cur[i] = cur[i] * dtau;
      for (int i = 0; i < nx; ++i) cur[i] = cur[i] + Lm[nx * (s + 1) + i];
    }
    for (int s = 0; s < dv_; ++s)  // :156-161
      Model::dHdu(&ret[nu * s], &X[nx * s], &U[nu * s], &ptau_[np * s], &Lm[nx * (s + 1)]);
  }

  // Forward-difference Jacobian-vector product — cgmres.hpp:164-175.
  // Uses the state left by control(): U, x_dxh, F_dxh_h, t.
  void Ax(T* out, const T* v) {
    const T h = T(tun_.h);
    std::vector<T>& Ub = ub_;  // the reference's stack array U_buf (cgmres.hpp:165)
    for (int i = 0; i < len_; ++i) Ub[i] = v[i] * h;        // :168
    for (int i = 0; i < len_; ++i) Ub[i] = Ub[i] + U_[i];   // :169
    F(out, Ub.data(), xh_.data(), t_ + h);                  // :170
    for (int i = 0; i < len_; ++i) out[i] = out[i] - Fh_[i];  // :173
    const T inv_h = T(1.0) / h;
    for (int i = 0; i < len_; ++i) out[i] = out[i] * inv_h;   // :174
  }

  // Matrix-free GMRES(k_max), warm start, no restart — gmres.hpp:28-112
  void gmres(T* x, const T* b) {
    const int L = len_, ld = kmax_ + 1;
    T* V = V_.data();
    T* H = H_.data();
    T* e = rho_.data();
    T* g = g_.data();
    n_ax_ = 0;
    k_used_ = 0;
    // :33-34 r0 = b - A x0
    Ax(&V[0], x);
    for (int i = 0; i < L; ++i) V[i] = b[i] - V[i];
    e[0] = nrm2(&V[0], L);  // :37
    if (e[0] < tol_) {      // :39-41
      exit_ = kExitSmallResidual;
      return;
    }
    scale_by_recip(&V[0], e[0], L);  // :44
    int k;
    exit_ = kExitNatural;
    for (k = 0; k < kmax_; ++k) {  // :46
      T* w = &V[L * (k + 1)];
      Ax(w, &V[L * k]);  // :48
      ++n_ax_;
      for (int i = 0; i <= k; ++i) {  // :52-58 modified Gram-Schmidt, three passes per basis vector
        const T* vi = &V[L * i];
        const T hik = dot(vi, w, L);
        H[ld * k + i] = hik;
        for (int l = 0; l < L; ++l) scratch_[l] = vi[l] * hik;
        for (int l = 0; l < L; ++l) w[l] = w[l] - scratch_[l];
      }
      const T hn = nrm2(w, L);  // :60
      H[ld * k + k + 1] = hn;
      if (std::fabs(hn) < T(DBL_EPSILON)) {  // :63-65 (DBL_EPSILON also in the float build: float.h constant)
        exit_ = kExitBreakdown;
        return;
      }
      scale_by_recip(w, hn, L);  // :67
      T* col = &H[ld * k];
      for (int i = 0; i < k; ++i) {  // :71-77 apply stored 2-vector reflectors
        const T* gi = &g[3 * i];
        const T beta = (gi[0] * col[i] + gi[1] * col[i + 1]) * gi[2];
        col[i] = col[i] - beta * gi[0];
        col[i + 1] = col[i + 1] - beta * gi[1];
      }
      // :80-85 new reflector; sign(0)=+1 (matrix.hpp:162); norm() of the 2-vector
      const T sigma = -(col[k] < T(0.0) ? T(-1.0) : T(1.0)) * nrm2(&col[k], 2);
      T* gk = &g[3 * k];
      gk[0] = col[k] - sigma;
      gk[1] = col[k + 1];
      gk[2] = T(2.0) / dot(gk, gk, 2);
      col[k] = sigma;
      col[k + 1] = T(0.0);
      // :88-90 rotate the residual vector
      const T beta = gk[0] * e[k] * gk[2];
      e[k] = e[k] - beta * gk[0];
      e[k + 1] = -beta * gk[1];
      if (std::fabs(e[k + 1]) < tol_) {  // :93-95 — k is NOT incremented on break
        exit_ = kExitConverged;
        break;
      }
    }
    k_used_ = k;
    // :100-107 back substitution on the leading k x k block (true division)
    for (int i = k - 1; i >= 0; --i) {
      for (int j = k - 1; j > i; --j) e[i] -= H[ld * j + i] * e[j];
      e[i] /= H[ld * i + i];
    }
    // :110-111 x += V[:,0:k] y ; accumulator is column k_max of V, zeroed first (matrix.hpp:82-91)
    T* acc = &V[L * kmax_];
    for (int l = 0; l < L; ++l) acc[l] = T(0.0);
    for (int j = 0; j < k; ++j)
      for (int l = 0; l < L; ++l) acc[l] += V[L * j + l] * e[j];
    for (int l = 0; l < L; ++l) x[l] = x[l] + acc[l];
  }

  // --- state access for teacher-forced tests -------------------------------------------
  T& t() { return t_; }
  std::vector<T>& U() { return U_; }
  std::vector<T>& dUdt() { return dUdt_; }
  std::vector<T>& Fh() { return Fh_; }
  std::vector<T>& xh() { return xh_; }
  std::vector<T>& V() { return V_; }
  std::vector<T>& H() { return H_; }
  std::vector<T>& rho() { return rho_; }
  std::vector<T>& g() { return g_; }
  std::vector<T>& ptau() { return ptau_; }
  int n_ax() const { return n_ax_; }      // Arnoldi mat-vecs executed inside the k loop
  int k_used() const { return k_used_; }  // size of the triangular solve
  int exit_reason() const { return exit_; }

  // matrix.hpp:166-224 — Gaussian elimination, partial pivoting, column-major, destroys mat & vec
  static void solve_in_place(T* vec, T* mat, int n) {
    for (int k = 0; k < n - 1; ++k) {
      int piv = k;
      T best = std::fabs(mat[n * k + k]);
      for (int i = k + 1; i < n; ++i)
        if (best < std::fabs(mat[n * k + i])) {
          best = std::fabs(mat[n * k + i]);
          piv = i;
        }
      if (piv != k) {
        T tmp = vec[k];
        vec[k] = vec[piv];
        vec[piv] = tmp;
        for (int j = k; j < n; ++j) {
          tmp = mat[n * j + k];
          mat[n * j + k] = mat[n * j + piv];
          mat[n * j + piv] = tmp;
        }
      }
      const T r = T(1.0) / mat[n * k + k];
      for (int i = k + 1; i < n; ++i) {
        mat[n * k + i] = mat[n * k + i] * r;
        for (int j = k + 1; j < n; ++j) mat[n * j + i] -= mat[n * k + i] * mat[n * j + k];
        vec[i] -= mat[n * k + i] * vec[k];
      }
    }
    for (int i = n - 1; i >= 0; --i) {
      for (int j = n - 1; j > i; --j) vec[i] -= mat[n * j + i] * vec[j];
      vec[i] /= mat[n * i + i];
    }
  }

 private:
  static T dot(const T* a, const T* b, int n) {  // matrix.hpp:151-159 sequential, index ascending
    T s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
  }
  static T nrm2(const T* a, int n) {  // matrix.hpp:140-148
    T s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * a[i];
    return std::sqrt(s);
  }
  static void scale_by_recip(T* a, T c, int n) {  // matrix.hpp:122-128
    const T r = T(1.0) / c;
    for (int i = 0; i < n; ++i) a[i] = a[i] * r;
  }

  int dv_, kmax_, len_;
  T tol_;
  Tuning tun_;
  T t_ = T(0);
  std::vector<T> U_, dUdt_, Fh_, xh_, ptau_;
  std::vector<T> V_, H_, rho_, g_, scratch_, xtau_, ltau_, b_, ub_;
  int n_ax_ = 0, k_used_ = 0, exit_ = kExitNatural;
};

}  // namespace oracle
