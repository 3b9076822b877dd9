"""Compare a regenerated fixture directory (FX_GOLDEN_OUT=<dir> make_golden.py) with the committed
tests/golden/*.npz: every array must be bit-identical (meta: equal after json parsing).
    python tests/golden/check_regen.py /tmp/regen"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main(other):
    bad = 0
    for fn in sorted(os.listdir(HERE)):
        if not fn.endswith(".npz"):
            continue
        path = os.path.join(other, fn)
        if not os.path.exists(path):
            print("MISSING", fn)
            bad += 1
            continue
        a, b = np.load(os.path.join(HERE, fn)), np.load(path)
        diffs = []
        if sorted(a.files) != sorted(b.files):
            diffs.append("keys %s" % sorted(set(a.files) ^ set(b.files)))
        for k in a.files:
            if k not in b.files:
                continue
            if k == "meta":
                ma, mb = (json.loads(bytes(z["meta"]).decode()) for z in (a, b))
                ma.pop("torch", None), mb.pop("torch", None)
                if ma != mb:
                    diffs.append("meta")
            elif a[k].shape != b[k].shape or not np.array_equal(a[k], b[k], equal_nan=True):
                d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() \
                    if a[k].shape == b[k].shape else "shape"
                diffs.append("%s (max |d| %s)" % (k, d))
        print("%-32s %s" % (fn, "identical" if not diffs else "DIFFERS: " + "; ".join(diffs[:6])))
        bad += bool(diffs)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1]) else 0)
