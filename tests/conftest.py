import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def ops():
    """The product op surface (HIP).  Import fails loudly if the .so is missing."""
    from vllm_amd import _custom_ops
    return _custom_ops


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(params=["balanced", "plain"])
def v1_form(request):
    """paged_attention_v1 has two launch forms: one workgroup per (sequence, kv head) (default)
    and balanced over the registered workspace ("pa_balance" = 1)."""
    from vllm_amd import _lib
    _lib.set_tuning("pa_balance", 0 if request.param == "plain" else 1)
    yield request.param
    _lib.set_tuning("pa_balance", 0)
