// Setup-side kernels: layout changes between the ABI's instance-major vectors and the device's
// element-major [n][ldb] arrays, the batched Newton initialisation (cgmres.hpp:61-76 with
// matrix.hpp:166-224 inside) and the registry probe.
#pragma once
#include <hip/hip_runtime.h>

#include "models.hip.h"

namespace cgm {

// dst[e*ldb + b] = src[(bcast ? 0 : b)*n + e]
template <class T>
__global__ void to_element_major(T* __restrict__ dst, const T* __restrict__ src, int B, int ldb, int n, int bcast) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (b < B) dst[size_t(e) * ldb + b] = src[size_t(bcast ? 0 : b) * n + e];
}
// dst[b*n + e] = src[e*ldb + b]
template <class T>
__global__ void to_instance_major(T* __restrict__ dst, const T* __restrict__ src, int B, int ldb, int n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;
  if (b < B) dst[size_t(b) * n + e] = src[size_t(e) * ldb + b];
}
// U[(s*nu + j)*ldb + b] = u0[(bcast?0:b)*nu + j] for every stage — cgmres.hpp:51-59
template <class T>
__global__ void replicate_stages(T* __restrict__ dst, const T* __restrict__ src, int B, int ldb, int n, int stages,
                                 int bcast) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = blockIdx.y;  // 0 .. n*stages-1
  if (b < B) dst[size_t(e) * ldb + b] = src[size_t(bcast ? 0 : b) * n + (e % n)];
}

// Gaussian elimination with partial pivoting on a column-major N x N system, in place — the
// statement order of matrix.hpp:166-224 (reciprocal pivot multiply in the elimination, true
// division in the back substitution).
template <class T, int N>
__device__ __forceinline__ void linsolve_dev(T* vec, T* mat) {
  for (int k = 0; k < N - 1; ++k) {
    int piv = k;
    T best = mat[N * k + k] < 0 ? -mat[N * k + k] : mat[N * k + k];
    for (int i = k + 1; i < N; ++i) {
      const T a = mat[N * k + i] < 0 ? -mat[N * k + i] : mat[N * k + i];
      if (best < a) {
        best = a;
        piv = i;
      }
    }
    if (piv != k) {
      T tmp = vec[k];
      vec[k] = vec[piv];
      vec[piv] = tmp;
      for (int j = k; j < N; ++j) {
        tmp = mat[N * j + k];
        mat[N * j + k] = mat[N * j + piv];
        mat[N * j + piv] = tmp;
      }
    }
    const T r = T(1.0) / mat[N * k + k];
    for (int i = k + 1; i < N; ++i) {
      mat[N * k + i] = mat[N * k + i] * r;
      for (int j = k + 1; j < N; ++j) mat[N * j + i] -= mat[N * k + i] * mat[N * j + k];
      vec[i] -= mat[N * k + i] * vec[k];
    }
  }
  for (int i = N - 1; i >= 0; --i) {
    for (int j = N - 1; j > i; --j) vec[i] -= mat[N * j + i] * vec[j];
    vec[i] /= mat[N * i + i];
  }
}

// init_u0_newton, cgmres.hpp:61-76: u0 <- u0 - (ddH/duu)^-1 dH/du, n_loop times, one lane per instance.
// All three arrays are instance-major; u0 is updated in place (the caller then replicates it over the stages).
template <class M, class T>
__global__ void newton_u0_kernel(T* __restrict__ u0, const T* __restrict__ x0, const T* __restrict__ p0, int B,
                                 int n_loop) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
  T x[NX], u[NU], p[NP > 0 ? NP : 1], l[NX], rhs[NU], mat[NU * NU], tr[M::NC > 0 ? M::NC : 1], f[NX];
  for (int i = 0; i < NX; ++i) x[i] = x0[size_t(b) * NX + i];
  for (int j = 0; j < NU; ++j) u[j] = u0[size_t(b) * NU + j];
  for (int j = 0; j < NP; ++j) p[j] = p0[size_t(b) * NP + j];
  typename M::Math mc;
  mc.init();
  M::dxdt(f, x, u, tr, mc);  // only for the trig values dHdu reads (they depend on x alone)
  M::dPhidx(l, x, p);
  for (int it = 0; it < n_loop; ++it) {
    M::dHdu(rhs, x, u, p, l, tr);
    M::ddHduu(mat, x, u, p, l);
    linsolve_dev<T, NU>(rhs, mat);
    for (int j = 0; j < NU; ++j) u[j] = u[j] - rhs[j];
  }
  for (int j = 0; j < NU; ++j) u0[size_t(b) * NU + j] = u[j];
}

// Registry probe: [dxdt | dPhidx | dHdx | dHdu] at one point, one thread.
template <class M>
__global__ void probe_kernel(const double* x, const double* u, const double* p, const double* l, double* out) {
  constexpr int NX = M::NX, NU = M::NU;
  double f[NX], g[NX], hx[NX], hu[NU], tr[M::NC > 0 ? M::NC : 1];
  typename M::Math mc;
  mc.init();
  M::dxdt(f, x, u, tr, mc);
  M::dPhidx(g, x, p);
  M::dHdx(hx, x, u, p, l, tr);
  M::dHdu(hu, x, u, p, l, tr);
  for (int i = 0; i < NX; ++i) out[i] = f[i], out[NX + i] = g[i], out[2 * NX + i] = hx[i];
  for (int j = 0; j < NU; ++j) out[3 * NX + j] = hu[j];
}

// Placement of an oversubscribed batch (WgParams::perm): instances grouped by the Arnoldi count of their last tick,
// the long-running ones FIRST (the hardware hands workgroups out in index order, so the expensive ones start early and
// the cheap ones fill the tail).  A STABLE counting sort of the B keys by one workgroup: inside a count group the
// instances keep their caller order (rank = number of earlier instances with the same key, from wave ballots and a
// per-wave count table), so the same counts always give the same placement — runs are reproducible.
// key: n_ax[b] in 0..kmax (gmres.hpp:93-95), clamped to 64 groups.
template <int UNUSED = 0>  // (a template: this header is included by several translation units)
__global__ __launch_bounds__(1024) void bin_by_count_kernel(int* __restrict__ perm, const int* __restrict__ n_ax, int B, int kmax) {
  constexpr int NB = 64, NW = 16;
  __shared__ int start[NB];      // next free slot of every group
  __shared__ int wcnt[NW][NB];   // per pass: members of group k in wave w
  const int nb = kmax + 1 < NB ? kmax + 1 : NB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  auto key_of = [&](int b) {
    int k = b < B ? n_ax[b] : -1;
    return k < 0 ? (b < B ? 0 : -1) : (k >= nb ? nb - 1 : k);
  };
  for (int q = tid; q < NB; q += blockDim.x) start[q] = 0;
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x) atomicAdd(&start[key_of(b)], 1);  // (counts only: order-free)
  __syncthreads();
  if (tid == 0) {  // start offsets, largest count first
    int off = 0;
    for (int k = nb - 1; k >= 0; --k) {
      const int c = start[k];
      start[k] = off;
      off += c;
    }
  }
  __syncthreads();
  for (int base = 0; base < B; base += blockDim.x) {  // 1024 consecutive instances per pass, in caller order
    const int b = base + tid, k = key_of(b);
    int rank = 0;
    for (int g = 0; g < nb; ++g) {
      const unsigned long long m = __ballot(k == g);
      if (k == g) rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wave][g] = __popcll(m);
    }
    __syncthreads();
    if (k >= 0) {
      int before = 0;
      for (int w = 0; w < wave; ++w) before += wcnt[w][k];
      perm[start[k] + before + rank] = b;
    }
    __syncthreads();
    if (tid < nb) {
      int tot = 0;
      for (int w = 0; w < NW; ++w) tot += wcnt[w][tid];
      start[tid] += tot;
    }
    __syncthreads();
  }
}

}  // namespace cgm
