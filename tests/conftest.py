import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_pkg():
    """Import the product package.  Its directory is named ``dash-infer_amd`` (not a
    valid Python identifier), so it is loaded under the module name ``dash_infer_amd``."""
    if "dash_infer_amd" in sys.modules:
        return sys.modules["dash_infer_amd"]
    path = os.path.join(ROOT, "dash-infer_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(
        "dash_infer_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dash_infer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
