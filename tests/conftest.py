"""Shared test plumbing: marker registration, golden-fixture loading, oracle import path."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True, scope="session")
def _route_hooks_enabled():
    """The tests reach every kernel route at small sizes through the library's test hooks (FOURQ_SPLIT_MIN, FOURQ_PAIR_MAX, ...),
    which the library reads only under FOURQ_DEBUG_ROUTES=1 (fourq_amd/csrc/fourq_amd.hip, tools/README.md)."""
    os.environ["FOURQ_DEBUG_ROUTES"] = "1"
    yield


def unhex(v):
    """Inverse of make_golden.hx: hex strings -> ints, lists -> tuples (recursively)."""
    if isinstance(v, bool):
        return v
    if isinstance(v, str):
        return int(v, 16)
    if isinstance(v, dict):
        return {k: (e if k.startswith("_") else unhex(e)) for k, e in v.items()}
    return tuple(unhex(e) for e in v)


def load_golden(name, raw=False):
    with open(os.path.join(GOLDEN, name)) as fh:
        data = json.load(fh)
    return data if raw else unhex(data)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name, raw=False):
        key = (name, raw)
        if key not in cache:
            cache[key] = load_golden(name, raw)
        return cache[key]

    return get


@pytest.fixture(scope="module", params=[False, True], ids=["select-by-address", "constant-time"])
def eng(request):
    """One engine per test module and selection mode: the default ladders (a digit of the scalar is a table address,
    as in the reference) and FOURQ_CT_SELECT's full-table scans must give bit-identical results on the whole suite."""
    from fourq_amd import Engine
    e = Engine(0)
    e.ct_select = request.param
    assert e.ct_select == request.param
    yield e
    e.close()
