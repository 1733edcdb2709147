import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The CPU oracle is OpenMP code.  A GPU box reports all of its host's cores (128+) but a one-GPU job owns a share of ~16: with the
# default team size every parallel region is oversubscribed 8x and, when the neighbours are busy, spin-waiting barriers turn a
# 1-second check into a minute (seen in round 2: a 70-second suite stalled for minutes).  Tests need the oracle's answers, not its
# speed: a team that fits the share, and sleeping waits.  (bench.py's cpu_baseline leg sets its own thread count.)
def _cpu_share():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_share()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg():
    """The product package (directory name fixed by the build contract, not a Python identifier)."""
    return importlib.import_module("thermodynamic-interpolation_amd")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def golden_weights(g):
    """Flat canonical weights of a painn golden case: either the synthetic ones (regenerated from the seed) or the
    reference-constructor state_dict stored in the file (sd::* keys)."""
    ti = pkg()
    variant, F, L = int(g["variant"]), int(g["F"]), int(g["L"])
    spec = ti.weights.painn_param_spec(variant, F, L, 25)
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")}
    if not sd:
        sd = ti.synthetic.painn_state_dict(variant, F, L, 25, int(g["seed"]))
        if "recipe_keys" in g:              # range fixtures: synthetic weights with some tensors rescaled
            sd = ti.synthetic.scale_state_dict(sd, list(zip([str(k) for k in g["recipe_keys"]], g["recipe_factors"])))
    return ti.weights.flatten_state_dict(sd, spec)


@pytest.fixture(scope="session")
def ti():
    return pkg()
