// A user's main in the style of <example>/main.cpp:35-73 for the user model of vdp_model.hpp: the unmodified
// facade (include/cgmres.hpp) finds the model's device build through CGMRES_HIP_MODEL_PLUGINS.
//   g++ -O2 -I<repo>/include -I<repo>/tests/user_models vdp_main.cpp -L<lib> -lcgmres_hip -o vdp_main
//   CGMRES_HIP_MODEL_PLUGINS=<repo>/cgmres_cpp_amd/lib/plugins/libcgmres_model_vdp.so ./vdp_main <instance> <ticks>
#include <stdio.h>
#include <stdlib.h>

#include "vdp_model.hpp"
using Model = VdpModel;
#include "cgmres.hpp"

int main(int argc, char** argv) {
  const int b = argc > 1 ? atoi(argv[1]) : 0, ticks = argc > 2 ? atoi(argv[2]) : 20;
  double x[2] = {1.0 + 0.1 * b, -0.5 + 0.05 * b}, p[2] = {0.2 * b, 0.05 * (b % 3)}, u0[3] = {0.1, 1.9, 0.03};
  double u[3], f[2], dx[2];
  Cgmres<Model> controller;
  controller.set_ptau_repeat(p);
  controller.init_u0(u0);
  controller.init_u0_newton(u0, x, p, 10);
  for (int t = 0; t < ticks; ++t) {
    controller.control(u, x);
    printf("%d %d %.17g %.17g %.17g %.17g %.17g\n", b, t, u[0], u[1], u[2], x[0], x[1]);
    Model::dxdt(f, x, u, p);
    mul(dx, f, Model::dt, 2);
    add(x, x, dx, 2);
  }
  return 0;
}
