import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_package():
    """Import the hyphenated package directory `hla-la_amd/` as module `hla_la_amd`."""
    if "hla_la_amd" in sys.modules:
        return sys.modules["hla_la_amd"]
    spec = importlib.util.spec_from_file_location(
        "hla_la_amd", os.path.join(ROOT, "hla-la_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "hla-la_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hla_la_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "hlala_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "hlala_gpu.h"))):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import oracle_binding
    return oracle_binding.Oracle
