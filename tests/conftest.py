import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()          # builds oracle/libpt_oracle.so on first use (gcc only)
    return oracle_py


@pytest.fixture(scope="session")
def api():
    from opencl_path_tracer_amd import api as _api
    return _api


@pytest.fixture(scope="session")
def cb_spec():
    from opencl_path_tracer_amd import scenes
    return scenes.cornell_box()


@pytest.fixture(scope="session")
def cb_oracle_scene(oracle, cb_spec):
    return oracle.load_scene(cb_spec)


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")
