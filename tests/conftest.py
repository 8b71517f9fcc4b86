import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")
    # PyTorch ships its own copy of the HIP runtime: it has to initialise BEFORE libmi355rt.so pulls in /opt/rocm's
    # (tests that hand torch buffers / streams to the library fail with "no ROCm-capable device" in the other order)
    try:
        import torch
        torch.cuda.is_available()
    except Exception:       # noqa: BLE001 — CPU-only runs do not need torch
        pass


def _ensure_built():
    pkg_lib = os.path.join(ROOT, "raytracer-rs_amd", "libmi355rt.so")
    orc_lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(pkg_lib) and os.path.exists(orc_lib)):
        ge.build()


@pytest.fixture(scope="session")
def pkg():
    _ensure_built()
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    return ge.load_oracle()


@pytest.fixture(scope="session")
def scene_io(pkg):
    import importlib
    return importlib.import_module("raytracer_rs_amd.scene_io")


@pytest.fixture(scope="session")
def scenes(scene_io):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = scene_io.load_scene_file(os.path.join(SCENES, name + ".scene"))
        return cache[name]
    return get


class Semantics:
    """One intersector semantics on both sides: flags for the HIP library and for the CPU oracle.
    reference_default: the reference's OctTreeIntersector (library: BVH + octree confirm step; oracle: its octree).
    octree_walk: the same semantics, the library walking the octree directly (slow cross-check path).
    true_closest: NoAccelerationIntersector semantics (library: BVH only; oracle: brute force)."""

    def __init__(self, name, gpu, orc):
        self.name, self.gpu, self.orc = name, gpu, orc

    def __repr__(self):
        return self.name


def _semantics(pkg, oracle, name):
    return {"reference_default": Semantics(name, 0, 0),
            "octree_walk": Semantics(name, pkg.FLAG_OCTREE_SEMANTICS, 0),
            "true_closest": Semantics(name, pkg.FLAG_TRUE_CLOSEST_HIT, oracle.FLAG_BRUTE_FORCE)}[name]


@pytest.fixture(params=["reference_default", "true_closest"])
def sem(request, pkg, oracle):
    """the two shipped semantics (default + opt-out)"""
    return _semantics(pkg, oracle, request.param)


@pytest.fixture(params=["reference_default", "octree_walk", "true_closest"])
def sem3(request, pkg, oracle):
    return _semantics(pkg, oracle, request.param)
