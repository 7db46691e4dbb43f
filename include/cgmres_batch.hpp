// cgmres_batch.hpp — `batch` controllers of one Model advancing in lock-step on one MI355X.
//
// The reference has no batching concept: multiple_controller/main.cpp:89-110 simply owns two Cgmres objects and
// calls them back to back.  CgmresBatch<Model> is the batched counterpart of Cgmres<Model> (cgmres.hpp): the same
// method names with a leading instance axis on every vector (instance-major, see include/cgmres_hip.h), plus
// device-pointer variants so a closed loop never has to leave HBM.
// CgmresBatchSharded<Model> owns one such batch per GPU of a node: contiguous shards (cgmres_hip_shard_bounds), one
// handle + stream per device, no exchange between the shards inside a tick (controllers are independent).
#pragma once
#include <memory>
#include <thread>
#include <vector>

#include "cgmres.hpp"

template <class Model>
class CgmresBatch {
 public:
  static constexpr uint16_t dim_x = Model::dim_x, dim_u = Model::dim_u, dim_p = Model::dim_p, dv = Model::dv;

  explicit CgmresBatch(int32_t batch, int32_t device = 0, void* hip_stream = nullptr) : batch_(batch) {
    cgmres_hip_config cfg = cgmres_detail::config_for<Model>(batch, device);
    cfg.stream = hip_stream;
    cgmres_detail::check(cgmres_hip_create(&cfg, &handle_), "create");
  }
  ~CgmresBatch() {
    if (handle_) cgmres_hip_destroy(handle_);
  }
  CgmresBatch(const CgmresBatch&) = delete;
  CgmresBatch& operator=(const CgmresBatch&) = delete;

  int32_t batch() const { return batch_; }
  cgmres_hip_handle native_handle() const { return handle_; }

  // per_instance = false: one vector shared by every instance; true: [batch][...]
  void set_ptau(const double* ptau, bool per_instance = true) {
    if (dim_p) cgmres_detail::check(cgmres_hip_set_ptau(handle_, ptau, per_instance), "set_ptau");
  }
  void set_ptau_repeat(const double* p, bool per_instance = true) {
    if (dim_p) cgmres_detail::check(cgmres_hip_set_ptau_repeat(handle_, p, per_instance), "set_ptau_repeat");
  }
  void init_u0(const double* u0, bool per_instance = true) {
    cgmres_detail::check(cgmres_hip_init_u0(handle_, u0, per_instance), "init_u0");
  }
  // u0 [batch][dim_u] in/out, x0 [batch][dim_x], p0 [batch][dim_p]
  void init_u0_newton(double* u0, const double* x0, const double* p0, uint16_t n_loop) {
    cgmres_detail::check(cgmres_hip_init_u0_newton(handle_, u0, x0, p0, n_loop), "init_u0_newton");
  }
  // u [batch][dim_u] out, x [batch][dim_x] in — host pointers, blocking
  void control(double* u, const double* x) { cgmres_detail::check(cgmres_hip_control(handle_, u, x), "control"); }
  // device pointers, asynchronous on the handle's stream
  void control_device(double* u_dev, const double* x_dev) {
    cgmres_detail::check(cgmres_hip_control_device(handle_, u_dev, x_dev), "control_device");
  }
  // n_ticks of { control; x += Simulator::dxdt(x,u)*dt } without leaving the GPU
  void closed_loop_device(double* x_dev, double* u_dev, int32_t n_ticks) {
    cgmres_detail::check(cgmres_hip_closed_loop_device(handle_, x_dev, u_dev, n_ticks), "closed_loop_device");
  }
  // the same loop with set_ptau (cgmres.hpp:36-39) before every tick: ptau_seq_dev [n_ticks][batch][dim_p*(dv+1)]
  // (per_instance) or [n_ticks][dim_p*(dv+1)] (broadcast), device pointer
  void closed_loop_device(double* x_dev, double* u_dev, int32_t n_ticks, const double* ptau_seq_dev, bool per_instance) {
    cgmres_detail::check(cgmres_hip_closed_loop_device_ptau(handle_, x_dev, u_dev, n_ticks, ptau_seq_dev, per_instance ? 1 : 0),
                         "closed_loop_device_ptau");
  }
  void synchronize() { cgmres_detail::check(cgmres_hip_synchronize(handle_), "synchronize"); }

  // Arnoldi mat-vecs executed and exit reason (CGMRES_HIP_EXIT_*) per instance for the last tick
  void status(std::vector<int32_t>* n_ax, std::vector<int32_t>* reason) const {
    n_ax->resize(batch_);
    reason->resize(batch_);
    cgmres_detail::check(cgmres_hip_get_status(handle_, n_ax->data(), reason->data()), "get_status");
  }

 private:
  int32_t batch_;
  cgmres_hip_handle handle_ = nullptr;
};


// `batch` controllers of one Model split over several GPUs of one node — the caller of multiple_controller/main.cpp:89-110
// owning its controllers on more than one device.  Shard r holds the instances [lo_r, hi_r) of cgmres_hip_shard_bounds
// (the rule of cgmres_cpp_amd/sharding.py); every host vector is [batch][...] in caller order and is cut along that
// rule.  A tick needs no exchange between shards, so there is no collective here: set-up calls and host-pointer
// control() fan out on one host thread per shard (each blocks on its own device), the device-resident closed loop is
// enqueued on every shard's stream and runs concurrently.  `devices` may name a device more than once (two shards on
// one card: how the tests run it).
template <class Model>
class CgmresBatchSharded {
 public:
  static constexpr uint16_t dim_x = Model::dim_x, dim_u = Model::dim_u, dim_p = Model::dim_p, dv = Model::dv;

  CgmresBatchSharded(int32_t batch, const std::vector<int32_t>& devices) : batch_(batch) {
    const int32_t world = int32_t(devices.size());
    if (world < 1 || batch < world) {
      fprintf(stderr, "CgmresBatchSharded: %d controllers on %d devices\n", batch, world);
      exit(-1);
    }
    for (int32_t r = 0; r < world; ++r) {
      Shard sh;
      cgmres_detail::check(cgmres_hip_shard_bounds(batch, world, r, &sh.lo, &sh.hi), "shard_bounds");
      sh.ctrl.reset(new CgmresBatch<Model>(sh.hi - sh.lo, devices[r]));
      const cgmres_hip_handle h = sh.ctrl->native_handle();
      cgmres_detail::check(cgmres_hip_malloc(h, reinterpret_cast<void**>(&sh.x_dev), sizeof(double) * (sh.hi - sh.lo) * dim_x), "malloc");
      cgmres_detail::check(cgmres_hip_malloc(h, reinterpret_cast<void**>(&sh.u_dev), sizeof(double) * (sh.hi - sh.lo) * dim_u), "malloc");
      shards_.push_back(std::move(sh));
    }
  }
  ~CgmresBatchSharded() {
    for (Shard& sh : shards_) {
      if (sh.x_dev) cgmres_hip_free(sh.ctrl->native_handle(), sh.x_dev);
      if (sh.u_dev) cgmres_hip_free(sh.ctrl->native_handle(), sh.u_dev);
    }
  }
  CgmresBatchSharded(const CgmresBatchSharded&) = delete;
  CgmresBatchSharded& operator=(const CgmresBatchSharded&) = delete;

  int32_t batch() const { return batch_; }
  int32_t shards() const { return int32_t(shards_.size()); }
  void bounds(int32_t r, int32_t* lo, int32_t* hi) const { *lo = shards_[r].lo, *hi = shards_[r].hi; }
  CgmresBatch<Model>& shard(int32_t r) { return *shards_[r].ctrl; }

  void set_ptau(const double* ptau, bool per_instance = true) {
    each([&](Shard& sh) { sh.ctrl->set_ptau(per_instance ? ptau + size_t(sh.lo) * dim_p * (dv + 1) : ptau, per_instance); });
  }
  void set_ptau_repeat(const double* p, bool per_instance = true) {
    each([&](Shard& sh) { sh.ctrl->set_ptau_repeat(per_instance ? p + size_t(sh.lo) * dim_p : p, per_instance); });
  }
  void init_u0(const double* u0, bool per_instance = true) {
    each([&](Shard& sh) { sh.ctrl->init_u0(per_instance ? u0 + size_t(sh.lo) * dim_u : u0, per_instance); });
  }
  void init_u0_newton(double* u0, const double* x0, const double* p0, uint16_t n_loop) {
    each([&](Shard& sh) {
      sh.ctrl->init_u0_newton(u0 + size_t(sh.lo) * dim_u, x0 + size_t(sh.lo) * dim_x, p0 ? p0 + size_t(sh.lo) * dim_p : nullptr, n_loop);
    });
  }
  // one tick of every controller through host pointers: u [batch][dim_u] out, x [batch][dim_x] in
  void control(double* u, const double* x) {
    each([&](Shard& sh) { sh.ctrl->control(u + size_t(sh.lo) * dim_u, x + size_t(sh.lo) * dim_x); });
  }
  // the closed loop of the example mains without leaving the GPUs: the plant state lives in every shard's HBM
  void upload_state(const double* x) {
    each([&](Shard& sh) {
      cgmres_detail::check(cgmres_hip_memcpy_h2d(sh.ctrl->native_handle(), sh.x_dev, x + size_t(sh.lo) * dim_x,
                                                 sizeof(double) * (sh.hi - sh.lo) * dim_x), "memcpy_h2d");
    });
  }
  void closed_loop_device(int32_t n_ticks) {  // asynchronous: every shard's ticks are enqueued on its own stream
    for (Shard& sh : shards_) sh.ctrl->closed_loop_device(sh.x_dev, sh.u_dev, n_ticks);
  }
  void synchronize() {
    for (Shard& sh : shards_) sh.ctrl->synchronize();
  }
  // gathers x [batch][dim_x] and the last tick's u [batch][dim_u] (either may be null) — the only traffic between the
  // shards and the caller
  void download_state(double* x, double* u) {
    each([&](Shard& sh) {
      const cgmres_hip_handle h = sh.ctrl->native_handle();
      if (x) cgmres_detail::check(cgmres_hip_memcpy_d2h(h, x + size_t(sh.lo) * dim_x, sh.x_dev, sizeof(double) * (sh.hi - sh.lo) * dim_x), "memcpy_d2h");
      if (u) cgmres_detail::check(cgmres_hip_memcpy_d2h(h, u + size_t(sh.lo) * dim_u, sh.u_dev, sizeof(double) * (sh.hi - sh.lo) * dim_u), "memcpy_d2h");
    });
  }
  void status(std::vector<int32_t>* n_ax, std::vector<int32_t>* reason) {
    n_ax->assign(batch_, 0);
    reason->assign(batch_, 0);
    for (Shard& sh : shards_) {
      std::vector<int32_t> a, b;
      sh.ctrl->status(&a, &b);
      std::copy(a.begin(), a.end(), n_ax->begin() + sh.lo);
      std::copy(b.begin(), b.end(), reason->begin() + sh.lo);
    }
  }

 private:
  struct Shard {
    int32_t lo = 0, hi = 0;
    std::unique_ptr<CgmresBatch<Model>> ctrl;
    double *x_dev = nullptr, *u_dev = nullptr;
  };
  template <class F>
  void each(F&& f) {  // one host thread per shard (every call below blocks on its own device)
    if (shards_.size() == 1) return f(shards_[0]);
    std::vector<std::thread> th;
    for (Shard& sh : shards_) th.emplace_back([&f, &sh] { f(sh); });
    for (std::thread& t : th) t.join();
  }
  int32_t batch_;
  std::vector<Shard> shards_;
};
