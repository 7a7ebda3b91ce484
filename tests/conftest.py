import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "spawns_gpu_children: starts child processes that use the GPU; runs before any test that "
                                       "initialises the GPU in this process (a process that has must not exec)")


def pytest_collection_modifyitems(config, items):
    first = [it for it in items if it.get_closest_marker("spawns_gpu_children")]
    rest = [it for it in items if not it.get_closest_marker("spawns_gpu_children")]
    items[:] = first + rest


@pytest.fixture(scope="session")
def golden():
    import helpers
    return helpers.load_golden()


@pytest.fixture(scope="session")
def oracle():
    import helpers
    return helpers.Oracle()


@pytest.fixture(scope="session")
def ica():
    import image_codecs_amd
    image_codecs_amd.build_library()
    return image_codecs_amd


@pytest.fixture(scope="session")
def gpu_ctx(ica):
    if not ica.gpu_available():
        pytest.fail("GPU test selected but the library sees no HIP device")
    ctx = ica.Context()
    yield ctx
    ctx.close()
