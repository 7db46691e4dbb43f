// Closed-loop example programs on the MI355X path, one binary:
//
//   closed_loop pendulum   [ticks] [out_prefix]   arm-type inverted pendulum  (shipped sizes dv=25, k_max=5)
//   closed_loop msd        [ticks] [out_prefix]   two-mass spring damper      (dv=50, k_max=5)
//   closed_loop semiactive [ticks] [out_prefix]   semi-active damper          (dv=50, k_max=5)
//   closed_loop multiple   [ticks] [out_prefix]   msd + pendulum stepped in one loop (two controllers)
//   closed_loop batch <B>  [ticks]                B perturbed pendulum controllers, dv=50, k_max=10, CgmresBatch
//   closed_loop sharded <B> <dev,dev,...> [ticks]  the same batch owned shard by shard on the listed devices
//                                                  (CgmresBatchSharded; a device may be listed twice), checked against
//                                                  one unsharded CgmresBatch on the first device
//
// The first four follow the reference's example mains step by step (<example>/main.cpp: initial state and guess
// :35-52, controller setup :54-57, loop :63-85 = control, forward-Euler plant via mul/add, one text line per tick)
// and write the same files: <prefix>_x.txt / <prefix>_u.txt, "%f" time then "\t%f" per component, and
// "Elapsed time = %f" (accumulated wall time of control() only) at the end.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <sys/time.h>

#include <string>
#include <vector>

#include "cgmres.hpp"
#include "cgmres_batch.hpp"
#include "matrix.hpp"
#include "models.hpp"

static double now_s() {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return double(tv.tv_sec) + double(tv.tv_usec) * 1e-6;
}

// one controller + its plant + its two trajectory files
template <class Model, class Plant>
struct Loop {
  Cgmres<Model> controller;
  std::vector<double> x, u, f;
  FILE *fx, *fu;
  Loop(const std::string& prefix, const double* x0, const double* u0, const double* p)
      : x(x0, x0 + Model::dim_x), u(u0, u0 + Model::dim_u), f(Model::dim_x, 0.0) {
    std::vector<double> pt(Model::dim_p * (Model::dv + 1) + 1, 0.0);
    for (int s = 0; s <= Model::dv; ++s)
      for (int j = 0; j < Model::dim_p; ++j) pt[Model::dim_p * s + j] = p[j];
    controller.set_ptau(pt.data());
    controller.init_u0(u.data());
    controller.init_u0_newton(u.data(), x.data(), pt.data(), 10);
    fx = fopen((prefix + "_x.txt").c_str(), "w");
    fu = fopen((prefix + "_u.txt").c_str(), "w");
    if (!fx || !fu) {
      perror("fopen");
      exit(1);
    }
  }
  ~Loop() {
    fclose(fx);
    fclose(fu);
  }
  void control() { controller.control(u.data(), x.data()); }
  void plant_and_log(int tick) {
    Plant::rhs(f.data(), x.data(), u.data());
    mul(f.data(), f.data(), Model::dt, Model::dim_x);  // x = x + dxdt * dt, as two passes like the reference
    add(x.data(), x.data(), f.data(), Model::dim_x);
    fprintf(fx, "%f", Model::dt * tick);
    fprintf(fu, "%f", Model::dt * tick);
    for (double v : x) fprintf(fx, "\t%f", v);
    for (double v : u) fprintf(fu, "\t%f", v);
    fprintf(fx, "\n");
    fprintf(fu, "\n");
  }
};

static const double kPi = 3.14159265358979;  // the literal of the example mains
static const double kPendX0[4] = {kPi, kPi, 0.0, 0.0}, kPendU0[3] = {0.0, 3.0, 0.01}, kPendP[2] = {kPi / 4.0, 0.0};
static const double kMsdX0[4] = {2.0, 2.0, 0.0, 0.0}, kMsdU0[6] = {0.0, 0.0, 10.0, 10.0, 5e-4, 5e-4}, kMsdP[2] = {1, -1};
static const double kSemiX0[2] = {2.0, 0.0}, kSemiU0[3] = {0.028393761456740, 0.166095020295846, 0.030103250483332};

template <class Model, class Plant>
static int run_single(const char* prefix, int ticks, const double* x0, const double* u0, const double* p) {
  Loop<Model, Plant> loop(prefix, x0, u0, p);
  double t_all = 0;
  for (int i = 0; i < ticks; ++i) {
    const double t0 = now_s();
    loop.control();
    t_all += now_s() - t0;
    loop.plant_and_log(i);
  }
  printf("Elapsed time = %f\n", t_all);
  return 0;
}

static int run_multiple(const char* prefix, int ticks) {  // multiple_controller/main.cpp:89-143
  using M1 = examples::MsdModel<50, 5>;
  using M2 = examples::PendulumModel<25, 5>;
  Loop<M1, examples::MsdPlant> l1(std::string(prefix) + "1", kMsdX0, kMsdU0, kMsdP);
  Loop<M2, examples::PendulumPlant> l2(std::string(prefix) + "2", kPendX0, kPendU0, kPendP);
  double t_all = 0;
  for (int i = 0; i < ticks; ++i) {
    const double t0 = now_s();
    l1.control();
    l2.control();
    t_all += now_s() - t0;
    l1.plant_and_log(i);
    l2.plant_and_log(i);
  }
  printf("Elapsed time = %f\n", t_all);
  return 0;
}

static double splitmix_u01(uint64_t* state) {
  uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return double(z >> 11) * (1.0 / 9007199254740992.0);
}

static int run_batch(int B, int ticks) {
  using M = examples::PendulumModel<50, 10>;
  CgmresBatch<M> ctrl(B);
  std::vector<double> x(size_t(B) * 4), u(size_t(B) * 3), p(size_t(B) * 2), f(4);
  uint64_t seed = 12345;
  for (int b = 0; b < B; ++b) {  // SURVEY.md §8(d) perturbation recipe
    double r[5];
    for (double& v : r) v = splitmix_u01(&seed);
    x[4 * b + 0] = kPi + 0.2 * (r[0] - 0.5), x[4 * b + 1] = kPi + 0.2 * (r[1] - 0.5);
    x[4 * b + 2] = 0.2 * (r[2] - 0.5), x[4 * b + 3] = 0.2 * (r[3] - 0.5);
    p[2 * b + 0] = (kPi / 4.0) * (0.5 + r[4]), p[2 * b + 1] = 0.0;
    for (int j = 0; j < 3; ++j) u[3 * b + j] = kPendU0[j];
  }
  ctrl.set_ptau_repeat(p.data());
  ctrl.init_u0(u.data());
  ctrl.init_u0_newton(u.data(), x.data(), p.data(), 10);
  const double t0 = now_s();
  for (int i = 0; i < ticks; ++i) {
    ctrl.control(u.data(), x.data());
    for (int b = 0; b < B; ++b) {
      examples::PendulumPlant::rhs(f.data(), &x[4 * b], &u[3 * b]);
      mul(f.data(), f.data(), M::dt, 4);
      add(&x[4 * b], &x[4 * b], f.data(), 4);
    }
  }
  const double secs = now_s() - t0;
  printf("batch %d x %d ticks through host pointers: %.3f s, %.0f control steps/s; u[0] = %.12f %.12f %.12f\n", B, ticks,
         secs, double(B) * ticks / secs, u[0], u[1], u[2]);
  return 0;
}

static void seeded_pendulum_batch(int B, std::vector<double>* x, std::vector<double>* u, std::vector<double>* p) {
  x->resize(size_t(B) * 4), u->resize(size_t(B) * 3), p->resize(size_t(B) * 2);
  uint64_t seed = 12345;
  for (int b = 0; b < B; ++b) {  // SURVEY.md §8(d) perturbation recipe
    double r[5];
    for (double& v : r) v = splitmix_u01(&seed);
    (*x)[4 * b + 0] = kPi + 0.2 * (r[0] - 0.5), (*x)[4 * b + 1] = kPi + 0.2 * (r[1] - 0.5);
    (*x)[4 * b + 2] = 0.2 * (r[2] - 0.5), (*x)[4 * b + 3] = 0.2 * (r[3] - 0.5);
    (*p)[2 * b + 0] = (kPi / 4.0) * (0.5 + r[4]), (*p)[2 * b + 1] = 0.0;
    for (int j = 0; j < 3; ++j) (*u)[3 * b + j] = kPendU0[j];
  }
}

// The caller of multiple_controller/main.cpp:89-110 with B controllers on several devices: set-up, then `ticks` of the
// closed loop resident on the GPUs, then ONE gather.  The same job on a single unsharded batch must give the same bits
// (controllers are independent and both sides take the mapping the library picks for the shard size).
static int run_sharded(int B, const char* devlist, int ticks) {
  using M = examples::PendulumModel<50, 10>;
  std::vector<int32_t> devices;
  for (const char* q = devlist; *q;) {
    devices.push_back(int32_t(strtol(q, const_cast<char**>(&q), 10)));
    if (*q == ',') ++q;
  }
  std::vector<double> x0, u0, p;
  seeded_pendulum_batch(B, &x0, &u0, &p);
  std::vector<double> xs(x0), us(u0), xr(x0), ur(u0);
  CgmresBatchSharded<M> sharded(B, devices);
  sharded.set_ptau_repeat(p.data());
  sharded.init_u0(us.data());
  sharded.init_u0_newton(us.data(), xs.data(), p.data(), 10);
  sharded.upload_state(xs.data());
  const double t0 = now_s();
  sharded.closed_loop_device(ticks);
  sharded.synchronize();
  const double secs = now_s() - t0;
  sharded.download_state(xs.data(), us.data());
  for (int r = 0; r < sharded.shards(); ++r) {
    int32_t lo, hi;
    sharded.bounds(r, &lo, &hi);
    printf("shard %d: device %d, controllers [%d, %d)\n", r, devices[r], lo, hi);
  }
  // reference run: one handle, shard by shard sizes do not matter for the bits of an instance
  double worst = 0.0;
  for (int r = 0; r < sharded.shards(); ++r) {
    int32_t lo, hi;
    sharded.bounds(r, &lo, &hi);
    CgmresBatch<M> one(hi - lo, devices[0]);
    one.set_ptau_repeat(&p[2 * lo]);
    one.init_u0(&ur[3 * lo]);
    one.init_u0_newton(&ur[3 * lo], &xr[4 * lo], &p[2 * lo], 10);
    double *xd, *ud;
    cgmres_detail::check(cgmres_hip_malloc(one.native_handle(), reinterpret_cast<void**>(&xd), sizeof(double) * 4 * (hi - lo)), "malloc");
    cgmres_detail::check(cgmres_hip_malloc(one.native_handle(), reinterpret_cast<void**>(&ud), sizeof(double) * 3 * (hi - lo)), "malloc");
    cgmres_detail::check(cgmres_hip_memcpy_h2d(one.native_handle(), xd, &xr[4 * lo], sizeof(double) * 4 * (hi - lo)), "h2d");
    one.closed_loop_device(xd, ud, ticks);
    one.synchronize();
    cgmres_detail::check(cgmres_hip_memcpy_d2h(one.native_handle(), &xr[4 * lo], xd, sizeof(double) * 4 * (hi - lo)), "d2h");
    cgmres_detail::check(cgmres_hip_memcpy_d2h(one.native_handle(), &ur[3 * lo], ud, sizeof(double) * 3 * (hi - lo)), "d2h");
    cgmres_hip_free(one.native_handle(), xd), cgmres_hip_free(one.native_handle(), ud);
  }
  for (size_t i = 0; i < xs.size(); ++i) worst = fmax(worst, fabs(xs[i] - xr[i]));
  for (size_t i = 0; i < us.size(); ++i) worst = fmax(worst, fabs(us[i] - ur[i]));
  printf("sharded %d controllers x %d ticks on %zu shard(s): %.3f s, %.0f control steps/s; max |sharded - single| = %.3g; "
         "u[0] = %.12f %.12f %.12f\n", B, ticks, devices.size(), secs, double(B) * ticks / secs, worst, us[0], us[1], us[2]);
  return worst == 0.0 ? 0 : 1;
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "pendulum";
  if (!strcmp(which, "batch")) return run_batch(argc > 2 ? atoi(argv[2]) : 4096, argc > 3 ? atoi(argv[3]) : 100);
  if (!strcmp(which, "sharded"))
    return run_sharded(argc > 2 ? atoi(argv[2]) : 4096, argc > 3 ? argv[3] : "0", argc > 4 ? atoi(argv[4]) : 100);
  const int ticks = argc > 2 ? atoi(argv[2]) : -1;
  const char* prefix = argc > 3 ? argv[3] : nullptr;
  if (!strcmp(which, "pendulum"))
    return run_single<examples::PendulumModel<25, 5>, examples::PendulumPlant>(
        prefix ? prefix : "arm_type_inverted_pendulum", ticks < 0 ? 10001 : ticks, kPendX0, kPendU0, kPendP);
  if (!strcmp(which, "msd"))
    return run_single<examples::MsdModel<50, 5>, examples::MsdPlant>(prefix ? prefix : "mass_spring_damper",
                                                                      ticks < 0 ? 20001 : ticks, kMsdX0, kMsdU0, kMsdP);
  if (!strcmp(which, "semiactive"))
    return run_single<examples::SemiactiveModel<50, 5>, examples::SemiactivePlant>(
        prefix ? prefix : "semiactive_damper", ticks < 0 ? 20001 : ticks, kSemiX0, kSemiU0, kSemiX0 /*unused: dim_p = 0*/);
  if (!strcmp(which, "multiple")) return run_multiple(prefix ? prefix : "multiple_controller_", ticks < 0 ? 10001 : ticks);
  fprintf(stderr, "usage: %s pendulum|msd|semiactive|multiple [ticks] [prefix]  |  batch <B> [ticks]  |  sharded <B> <dev,dev,..> [ticks]\n", argv[0]);
  return 2;
}
