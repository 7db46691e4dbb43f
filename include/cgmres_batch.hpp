// cgmres_batch.hpp — `batch` controllers of one Model advancing in lock-step on one MI355X.
//
// The reference has no batching concept: multiple_controller/main.cpp:89-110 simply owns two Cgmres objects and
// calls them back to back.  CgmresBatch<Model> is the batched counterpart of Cgmres<Model> (cgmres.hpp): the same
// method names with a leading instance axis on every vector (instance-major, see include/cgmres_hip.h), plus
// device-pointer variants so a closed loop never has to leave HBM.
#pragma once
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
