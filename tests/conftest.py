import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


class Golden:
    """Lazy access to the committed fixtures (tests/golden/*.npz + manifest.json)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, 'manifest.json')) as fh:
            self.manifest = json.load(fh)
        self.cases = {c["id"]: c for c in self.manifest["cases"]}
        self._npz = {}

    def arrays(self, case):
        grp = case["group"]
        if grp not in self._npz:
            self._npz[grp] = np.load(os.path.join(GOLDEN, grp + '.npz'))
        z = self._npz[grp]
        return ({k: z[v] for k, v in case["in"].items()}, {k: z[v] for k, v in case["out"].items()})

    def ids(self, *ops, group=None):
        return [c["id"] for c in self.manifest["cases"] if (not ops or c["op"] in ops) and (group is None or c["group"] == group)]


_GOLDEN = Golden()


@pytest.fixture(scope="session")
def golden():
    return _GOLDEN


def golden_ids(*ops, group=None):
    return _GOLDEN.ids(*ops, group=group)


@pytest.fixture
def oracle_native(monkeypatch):
    """Route the package's three native primitives to the CPU oracle (host-logic tests only)."""
    import oracle_backend
    from oflibpytorch_amd import _native
    for name in ("flow_flags", "warp_bwd", "splat_fwd", "device", "sample_pts", "flow_extents", "warp_bwd_win", "splat_fwd_win",
                 "_wants_grad", "flow_from_matrix", "warp_valid"):
        monkeypatch.setattr(_native, name, getattr(oracle_backend, name))
    return oracle_backend
