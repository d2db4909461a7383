"""pytest wiring: `gpu` marker, import paths, golden-fixture loader."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "prot2text-v2-esm3_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    d = {k: z[k] for k in z.files}
    if "meta_json" in d:
        d["meta"] = json.loads(bytes(d.pop("meta_json")).decode())
    return d


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
