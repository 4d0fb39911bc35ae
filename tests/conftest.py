import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """Measured parity margins of this run (tests/parity.py): gpurun_out/ travels back from the GPU box."""
    try:
        import parity
        parity.dump_margins(os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    except Exception as e:                      # never turn a green run red over the report
        print("parity margins not written:", e)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def olib():
    from oracle import oracle
    return oracle.lib()


def angle_close(a, b, tol):
    """compare angles modulo 2*pi"""
    d = np.abs(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) + np.pi) % (2 * np.pi) - np.pi)
    return np.max(d) if d.size else 0.0


def shove_ants_into_box(o, rng):
    """Puts every ant of the oracle engine `o` against the +x face of its box, moving towards it: the ant-box contact
    paths (narrow phase, rank-1 fold, reaction wrench on the box) are exercised from the next step on."""
    A = o.num_agents
    roots = o.tensor("root_states").reshape(o.num_envs, A + 1, 13)
    box = roots[:, A, :]
    half_x = 0.5
    for k in range(A):
        roots[:, k, 0] = box[:, 0] + half_x + rng.uniform(0.20, 0.45, o.num_envs).astype(np.float32)   # torso sphere r = 0.25: touching / overlapping
        roots[:, k, 2] = rng.uniform(0.45, 0.75, o.num_envs).astype(np.float32)
        roots[:, k, 7] = rng.uniform(-2.5, -0.5, o.num_envs).astype(np.float32)                         # vx towards the box


def random_dr_params(rng, num_ants):
    """[num_ants, 33] physical domain-randomisation blocks drawn from the ranges of cfg/TenAnt.yaml:97-122: mass and damping
    scales U(0.5, 1.5), joint-limit offsets N(0, 0.01) (include/mms.h: mms_set_dr)."""
    dr = np.ones((num_ants, 33), np.float32)
    dr[:, 0:9] = rng.uniform(0.5, 1.5, (num_ants, 9))
    dr[:, 9:17] = rng.uniform(0.5, 1.5, (num_ants, 8))
    dr[:, 17:33] = rng.normal(0.0, 0.01, (num_ants, 16))
    return dr
