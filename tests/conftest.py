import pathlib
import sys

import pytest

import os

ROOT = pathlib.Path(__file__).resolve().parents[1]
# tests switch kernel variants / thresholds through PGX_* variables (monkeypatch.setenv): the Python loader copies them into the
# library's tuning table (include/pgx.h: pgx_tuning_set) only with this opt-in - libpgx.so itself never reads them
os.environ.setdefault("PGX_TUNING_FROM_ENV", "1")
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def require_gpu():
    if not _gpu_available():
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback)")
