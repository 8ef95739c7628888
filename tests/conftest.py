"""Test configuration.

Markers: ``gpu`` tests need a real MI355X and call the product through its C ABI;
everything else runs on CPU (oracle vs golden vectors, host logic, ABI surface).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should fail loudly, not skip: nothing to do here.
    pass


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.load()
    return oracle_py


@pytest.fixture(scope="session")
def mesh_dir(tmp_path_factory):
    """Small 2-level m6wing-style input in the reference's file formats (722 / 124 nodes)."""
    from mgcfd import meshgen
    d = tmp_path_factory.mktemp("mesh_m6_small")
    mg = meshgen.make_multigrid((9, 5), "m6wing", seed=3, cavity_radius=0.15, jitter=0.2,
                                area_noise=0.05, volume_noise=0.05)
    meshgen.write_input(mg, str(d))
    return str(d)


@pytest.fixture(scope="session")
def mesh3_dir(tmp_path_factory):
    """3-level input with non-nested lattices (13/9/6) so few nodes coincide with their parents."""
    from mgcfd import meshgen
    d = tmp_path_factory.mktemp("mesh_m6_3lvl")
    mg = meshgen.make_multigrid((13, 9, 6), "m6wing", seed=11, cavity_radius=0.12, jitter=0.25,
                                area_noise=0.08, volume_noise=0.1)
    meshgen.write_input(mg, str(d))
    return str(d)


@pytest.fixture(scope="session")
def fvcorr_dir(tmp_path_factory):
    """Single-level fvcorr-style input: box minus its centre node (SURVEY.md §8d cfg1 recipe)."""
    from mgcfd import meshgen
    d = tmp_path_factory.mktemp("mesh_fvcorr_small")
    mg = meshgen.make_multigrid((12,), "fvcorr", seed=5, cavity_radius=0.01, volume_noise=0.02)
    meshgen.write_input(mg, str(d))
    return str(d)


def perturbed_state(nel, ff_var, seed, amplitude=0.01):
    """Far-field state with uniform(-amplitude, +amplitude) relative noise."""
    rng = np.random.default_rng(seed)
    base = np.tile(np.asarray(ff_var, dtype=np.float64), (nel, 1))
    base[:, 2:4] += 0.3            # momentum y/z are exactly 0 at far field: make them non-trivial
    return base * (1.0 + amplitude * rng.uniform(-1.0, 1.0, base.shape))
