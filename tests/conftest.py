import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The checker module (oracle/orc.py), with liboracle.so built on demand (gcc only, no GPU)."""
    from oracle import orc as _orc
    if not os.path.exists(_orc.ORACLE_SO):
        _orc.build(ref=os.path.isdir(_orc.REFERENCE_ROOT))
    return _orc


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_ids():
    return [os.path.basename(p)[:-4] for p in golden_files()]


def load_golden(path):
    z = np.load(path)
    g = {k: z[k] for k in z.files}
    model, dv, kmax, nx, nu, npar, is32 = [int(v) for v in g["meta"]]
    g["_case"] = dict(model=model, dv=dv, kmax=kmax, dim_x=nx, dim_u=nu, dim_p=npar,
                      dtype="f32" if is32 else "f64", tol=float(g["tol"][0]))
    g["_ticks"] = sorted(int(k[4:-2]) for k in g if k.startswith("tick") and k.endswith("_t"))
    return g
