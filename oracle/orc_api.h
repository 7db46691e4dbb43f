/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * One C ABI shared by the two checker libraries so a single ctypes wrapper (oracle/orc.py) drives both:
 *   oracle/liboracle.so       — the repo's CPU restatement (oracle/cgmres_oracle.hpp)
 *   oracle/_ref/libref.so     — the UNMODIFIED reference headers compiled in place from /root/reference
 * All vectors cross the ABI as double (fp32 instances convert at the boundary), instance layout is the
 * reference's own: U[dim_u*stage + j], ptau[dim_p*stage + j], V column-major L x (k_max+1),
 * H column-major with leading dimension k_max+1.
 */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

/* model: 0 pendulum, 1 mass-spring-damper, 2 semiactive damper.  dtype: 0 fp64, 1 fp32.
 * tol < 0 selects the model's shipped tol (1e-6).  Returns NULL when the combination is not built
 * (libref only instantiates the table in oracle/ref_harness.cpp). */
void* orc_create(int model, int dv, int kmax, double tol, int dtype);
void orc_destroy(void* c);
/* out[0..6] = dim_x, dim_u, dim_p, dv, kmax, len, dtype */
void orc_dims(void* c, int* out);
/* out[0..4] = dt, h, zeta, Tf, alpha */
void orc_tuning(void* c, double* out);

void orc_set_ptau(void* c, const double* ptau);          /* dim_p*(dv+1) */
void orc_init_u0(void* c, const double* u0);             /* dim_u */
void orc_init_u0_newton(void* c, double* u0, const double* x0, const double* p0, int n_loop);
void orc_control(void* c, double* u, const double* x);   /* one tick */

void orc_get_state(void* c, double* t, double* U, double* dUdt);
void orc_set_state(void* c, double t, const double* U, const double* dUdt);

/* white-box hooks */
void orc_F(void* c, double* ret, const double* U, const double* x, double t);
/* control()'s preamble only: sets x_dxh and F_dxh_h from the current U,t and the given x, returns b */
void orc_prepare(void* c, double* b, const double* x);
void orc_Ax(void* c, double* out, const double* v);       /* needs orc_prepare or orc_control first */
void orc_gmres(void* c, double* x_inout, const double* b); /* needs orc_prepare first */
/* sizes: V len*(kmax+1), H (kmax+1)^2, rho kmax+1, g 3*kmax; any pointer may be NULL */
void orc_get_krylov(void* c, double* V, double* H, double* rho, double* g);
/* out[0] = Arnoldi mat-vecs executed in the k loop of the last gmres; out[1] = triangular-solve size
 * (-1 when the library cannot observe it); out[2] = exit reason (-1 when unknown) */
void orc_last_solve(void* c, int* out);
/* plant right-hand side of the example's simulator.hpp */
void orc_plant(void* c, double* dxdt, const double* x, const double* u);

/* CPU-baseline driver (bench.py cpu_baseline leg): `ticks` closed-loop ticks (control, then the example's
 * forward-Euler plant step, <example>/main.cpp:66-73) of n independent controllers, statically partitioned
 * over nthreads std::threads.  x [n][dim_x] in/out, u [n][dim_u] out (last tick).  Returns wall seconds. */
double orc_run_closed_loop(void** ctrls, int n, double* x, double* u, int ticks, int nthreads);

#ifdef __cplusplus
}
#endif
