import os
import sys

import pytest

# torch ships its own HIP/HSA runtime libraries.  They must be the first ones loaded in a process
# that also loads librt_amd.so (linked against the system ROCm): loaded second, torch finds no GPU.
try:
    import torch  # noqa: F401
except ImportError:  # the CPU-only checks that do not need torch still run
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SCENES_DIR = os.path.join(ROOT, "tests", "golden", "scenes")
ALL_SCENES = ["scene1", "scene2", "scene3", "scene4-diffuse", "scene4-emission", "scene4-specular", "scene5", "scene6", "scene7"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(SCENES_DIR, name + ".test")


@pytest.fixture(scope="session")
def oracle():
    import checkers
    return checkers.load_oracle()


@pytest.fixture(scope="session")
def gpu():
    """The product library initialised on cuda:0 / HIP device 0."""
    from raytracer_amd import api
    api.init(0)
    return api
