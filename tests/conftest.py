import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import _sanafe_pkg  # noqa: E402

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


@pytest.fixture(scope="session")
def S():
    return _sanafe_pkg.load()


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "sanafe_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    yield


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "src"))


def ref_models_lib():
    return os.path.join(ROOT, "oracle", "_ref", "libsanafe_ref_models.so")
