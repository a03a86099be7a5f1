import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process GPU tests (tests/test_gpu_two_ranks.py) fork their ranks from a FORK SERVER that must exist before
    # anything in this process initialises HIP: a process that has touched the GPU must not exec (the pool refuses it) and
    # its forked copies cannot use the GPU.  Starting the server costs one exec of a small Python process, here and now.
    import multiprocessing
    import multiprocessing.forkserver
    try:
        multiprocessing.get_context("forkserver")
        multiprocessing.forkserver.ensure_running()
    except Exception as e:          # (platforms without forkserver: the tests that need it are skipped)
        config._gatres_forkserver_error = e


@pytest.fixture(scope="session")
def fork_ctx(request):
    err = getattr(request.config, "_gatres_forkserver_error", None)
    if err is not None:
        pytest.skip(f"no fork server: {err}")
    import multiprocessing
    return multiprocessing.get_context("forkserver")


@pytest.fixture(scope="session")
def pkg():
    import gnn_pressure_estimation_amd as G
    return G


@pytest.fixture(scope="session")
def oracle():
    from oracle import gatres_oracle
    return gatres_oracle


@pytest.fixture(scope="session")
def lib(pkg):
    return pkg._native.load()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, plus: the library reads its GATRES_* switches from the environment ONCE
    (csrc/gatres_common.h: gatres_knobs), so every change of such a variable is followed by gatres_knobs_reload(),
    and once more when the test's environment has been restored."""
    import gnn_pressure_estimation_amd as G

    def reload():
        G._native.load().gatres_knobs_reload()

    class _Env:
        def __getattr__(self, name):
            return getattr(monkeypatch, name)

        def setenv(self, name, value, *a, **k):
            monkeypatch.setenv(name, value, *a, **k)
            if name.startswith("GATRES_"):
                reload()

        def delenv(self, name, *a, **k):
            monkeypatch.delenv(name, *a, **k)
            if name.startswith("GATRES_"):
                reload()

    yield _Env()
    monkeypatch.undo()
    reload()
