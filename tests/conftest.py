import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden(object):
    """One tests/golden/<name>.npz produced by make_golden.py from the real reference."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.spec = self.meta["spec"]
        self.state0, self.state1, self.expect, batches = {}, {}, {}, {}
        for k in z.files:
            if k.startswith("state0/"):
                self.state0[k[7:]] = z[k]
            elif k.startswith("state1/"):
                self.state1[k[7:]] = z[k]
            elif k.startswith("expect/"):
                self.expect[k[7:]] = z[k]
            elif k.startswith("batch"):
                i, f = k.split("/", 1)
                batches.setdefault(int(i[5:]), {})[f] = z[k]
        self.batches = [batches[i] for i in sorted(batches)]

    @property
    def features(self):
        from collections import OrderedDict
        return OrderedDict((k, v) for item in self.spec["features"] for k, v in item.items())

    def cfg(self):
        m = self.meta
        return {"model": m["model"], "n_hidden": len(m["hidden"]), "n_cross": m.get("n_cross", 0)}


GOLDEN_CASES = ["deepfm_adam", "deepfm_adam_clip", "deepfm_sgd", "deepfm_d10", "dcnv2_adam"]


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return Golden(request.param)
