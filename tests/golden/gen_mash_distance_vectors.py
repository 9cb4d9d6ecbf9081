"""Generates tests/golden/mash_distance_vectors.json (+ euclidean vectors).

Runs ONLY in the build container, where /root/reference exists: it imports the
reference's genuine pure-Python ``diverse_seq/distance.py`` (third-party and
compiled modules replaced by empty stubs, none of which the two functions used
here touch) and records input -> output vectors for
``mash_distance`` (distance.py:230-291) and ``euclidean_distance``
(distance.py:335-336).  The JSON is data; nothing from the reference travels.

    python tests/golden/gen_mash_distance_vectors.py
"""
import importlib.util
import json
import pathlib
import sys
import types

import numpy as np

REF = pathlib.Path("/root/reference/diverse_seq/distance.py")
OUT = pathlib.Path(__file__).with_name("mash_distance_vectors.json")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference_distance():
    ident = lambda *a, **k: (lambda f: f) if not (a and callable(a[0])) else a[0]
    _stub("cogent3")
    _stub("cogent3.app")
    _stub("cogent3.app.typing", SeqsCollectionType=object, PairwiseDistanceType=object)
    sys.modules["cogent3"].app = sys.modules["cogent3.app"]
    sys.modules["cogent3.app"].typing = sys.modules["cogent3.app.typing"]
    _stub("cogent3.evolve")
    _stub("cogent3.evolve.fast_distance", DistanceMatrix=object)
    _stub("scinexus")
    _stub("scinexus.composable", define_app=ident)
    _stub("scinexus.progress", Progress=object)
    pkg = _stub("diverse_seq")
    pkg.__path__ = []
    _stub("diverse_seq._dvs", LazySeq=object)
    _stub("diverse_seq.util", _get_canonical_states=None, populate_inmem_zstore=None)
    pkg._dvs = sys.modules["diverse_seq._dvs"]
    pkg.util = sys.modules["diverse_seq.util"]
    spec = importlib.util.spec_from_file_location("diverse_seq.distance", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_distance()
    rng = np.random.default_rng(20260421)
    cases = []

    def add(left, right, k, s):
        left = sorted(set(int(x) for x in left))
        right = sorted(set(int(x) for x in right))
        try:
            d = ref.mash_distance(left, right, k, s)
        except ZeroDivisionError:
            d = "ZeroDivisionError"
        cases.append({"left": left, "right": right, "k": k, "sketch_size": s, "distance": d})

    add([1, 2, 3, 5], [2, 3, 4, 9], 12, 4)
    add([1, 2, 3], [1, 2, 3], 6, 3)
    add([1, 2, 3], [4, 5, 6], 6, 3)
    add([], [], 6, 3)
    add([], [1, 2], 6, 3)
    add([7], [7], 1, 1)
    add([1, 9], [9], 2, 5)
    for _ in range(120):
        s = int(rng.choice([1, 2, 5, 16, 50, 400]))
        k = int(rng.integers(1, 21))
        nl = int(rng.integers(0, s + 1)) if rng.random() < 0.4 else s
        nr = int(rng.integers(0, s + 1)) if rng.random() < 0.4 else s
        hi = int(rng.choice([s + 2, 3 * s + 3, 2**32 - 1]))
        left = rng.choice(hi, size=min(nl, hi), replace=False)
        right = rng.choice(hi, size=min(nr, hi), replace=False)
        if rng.random() < 0.3 and left.size and right.size:  # force overlap
            m = min(left.size, right.size) // 2
            right[:m] = left[:m]
        add(left, right, k, s)

    eu = []
    for _ in range(10):
        n = int(rng.choice([4, 16, 256]))
        a = rng.random(n); a /= a.sum()
        b = rng.random(n); b /= b.sum()
        eu.append({"a": a.tolist(), "b": b.tolist(),
                   "distance": float(ref.euclidean_distance(a, b))})
    OUT.write_text(json.dumps({"source": "diverse_seq/distance.py:230-291,335-336",
                               "mash_distance": cases, "euclidean_distance": eu}))
    print(f"wrote {len(cases)} mash + {len(eu)} euclidean vectors to {OUT}")


if __name__ == "__main__":
    main()
