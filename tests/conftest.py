import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _library_sees_the_restored_environment():
    """libmdt_hip reads its MDT_* switches once; a test that changed them (monkeypatch) tells the library, and after the
    test — when monkeypatch has put the environment back — the library re-reads it."""
    yield
    try:
        from multimodaldiscussiontransformer_amd import _lib
        _lib.reload_env()
    except Exception:  # noqa: BLE001  (library not built: CPU-only collection)
        pass
