"""Initial conditions of the reference's example programs and the seeded per-instance perturbation
used for batched runs:
  arm_type_inverted_pendulum/main.cpp:35-52, mass_spring_damper/main.cpp:35-55, semiactive_damper/main.cpp:35-40.
"""
import numpy as np

PI_TILDE = 3.14159265358979  # the literal the example mains use
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64_u01(seed, n):
    """n draws of u01 = (z >> 11) * 2**-53 from splitmix64(seed), vectorised."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def shipped(model):
    """(x0, u0 guess, p) exactly as the example main sets them."""
    if model in (0, "pendulum"):
        return (np.array([PI_TILDE, PI_TILDE, 0.0, 0.0]), np.array([0.0, 3.0, 0.01]),
                np.array([PI_TILDE / 4.0, 0.0]))
    if model in (1, "msd"):
        return (np.array([2.0, 2.0, 0.0, 0.0]), np.array([0.0, 0.0, 10.0, 10.0, 5e-4, 5e-4]), np.array([1.0, -1.0]))
    if model in (2, "semiactive"):
        return (np.array([2.0, 0.0]), np.array([0.028393761456740, 0.166095020295846, 0.030103250483332]),
                np.zeros(0))
    raise ValueError(model)


def batch(model, n, seed=12345):
    """Perturbed (x0 [n,dim_x], u0 guess [n,dim_u], p [n,dim_p]); draws per instance in order r1, r2, ..."""
    x0, u0, p = shipped(model)
    if model in (0, "pendulum"):
        r = splitmix64_u01(seed, 5 * n).reshape(n, 5)
        x = np.stack([PI_TILDE + 0.2 * (r[:, 0] - 0.5), PI_TILDE + 0.2 * (r[:, 1] - 0.5),
                      0.2 * (r[:, 2] - 0.5), 0.2 * (r[:, 3] - 0.5)], axis=1)
        pp = np.stack([(PI_TILDE / 4.0) * (0.5 + r[:, 4]), np.zeros(n)], axis=1)
    elif model in (1, "msd"):
        r = splitmix64_u01(seed, 2 * n).reshape(n, 2)
        x = np.stack([2.0 + 0.4 * (r[:, 0] - 0.5), 2.0 + 0.4 * (r[:, 1] - 0.5), np.zeros(n), np.zeros(n)], axis=1)
        pp = np.tile(p, (n, 1))
    else:
        r = splitmix64_u01(seed, n).reshape(n, 1)
        x = np.stack([2.0 + 0.4 * (r[:, 0] - 0.5), np.zeros(n)], axis=1)
        pp = np.zeros((n, 0))
    return np.ascontiguousarray(x), np.tile(u0, (n, 1)), np.ascontiguousarray(pp)
