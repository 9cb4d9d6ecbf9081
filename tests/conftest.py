import json
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rust_vectors():
    return json.loads((GOLDEN / "rust_unit_vectors.json").read_text())


@pytest.fixture(scope="session")
def mash_vectors():
    return json.loads((GOLDEN / "mash_distance_vectors.json").read_text())


# cogent3 DNA alphabet order T,C,A,G (diverse_seq/util.py:41-45; tests/test_util.py:9-16)
_DNA = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("TCAG"):
    _DNA[ord(_c)] = _i
    _DNA[ord(_c.lower())] = _i
_DNA[ord("U")] = 0
_DNA[ord("u")] = 0


def str2arr(seq: str) -> np.ndarray:
    return _DNA[np.frombuffer(seq.encode(), dtype=np.uint8)]


def read_fasta(path) -> dict:
    seqs, name, chunks = {}, None, []
    for line in pathlib.Path(path).read_text().splitlines():
        if line.startswith(">"):
            if name is not None:
                seqs[name] = "".join(chunks)
            name, chunks = line[1:].strip().split()[0], []
        elif line.strip():
            chunks.append(line.strip())
    if name is not None:
        seqs[name] = "".join(chunks)
    return seqs


@pytest.fixture(scope="session")
def brca1():
    """{name: uint8 codes} of the degapped BRCA1 demo alignment (config C1)"""
    raw = read_fasta(GOLDEN / "brca1.fasta")
    return {n: str2arr(s.replace("-", "").replace("?", "")) for n, s in raw.items()}


def synth_seqs(nseq, length, seed, invalid_frac=0.0, ragged=False):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(nseq):
        n = int(rng.integers(max(1, length // 2), length + 1)) if ragged else length
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        if invalid_frac:
            s[rng.random(n) < invalid_frac] = 4
        out.append(s)
    return out


def pack_reference(src: np.ndarray):
    """numpy statement of the packed sequence form (include/dvs_hip.h "packed sequences"): per 16 bases a uint32
    of 2-bit codes with the FIRST base in bits 31..30 and a uint16 whose bit 15 - i flags base i as invalid
    (>= 4, or behind the end)"""
    n = src.size
    nw = (n + 15) // 16
    pad = np.full(nw * 16, 255, np.uint8)
    pad[:n] = src
    g = pad.reshape(nw, 16).astype(np.uint32)
    codes = ((g & 3) << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
    mask = ((g > 3).astype(np.uint32) << (15 - np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint16)
    return codes, mask


def clades(newick_or_tuple) -> set[frozenset]:
    """the set of leaf sets below every internal node (for topology comparison)"""
    def parse(text: str):
        text = text.strip().rstrip(";")
        pos = 0

        def node():
            nonlocal pos
            while text[pos] == " ":
                pos += 1
            if text[pos] == "(":
                pos += 1
                kids = [node()]
                while text[pos] == ",":
                    pos += 1
                    kids.append(node())
                assert text[pos] == ")"
                pos += 1
                while pos < len(text) and text[pos] == " ":
                    pos += 1
                return tuple(kids)
            start = pos
            while pos < len(text) and text[pos] not in ",()":
                pos += 1
            return text[start:pos].strip()

        return node()

    tree = parse(newick_or_tuple) if isinstance(newick_or_tuple, str) else newick_or_tuple
    out: set[frozenset] = set()

    def walk(t):
        if isinstance(t, str):
            return frozenset([t])
        leaves = frozenset().union(*[walk(c) for c in t])
        out.add(leaves)
        return leaves

    walk(tree)
    return out


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment(monkeypatch):
    """The library reads its DVS_* switches once per context (dvs_ctx_create).  Tests flip them with
    monkeypatch.setenv / delenv in the middle of a module-scoped context's life: every such change, and its
    undoing at the end of the test, is followed by a re-read in every live context."""
    def refresh(name):
        if str(name).startswith(("DVS_", "HSA_CU_MASK", "ROC_GLOBAL_CU_MASK")):
            try:
                from diverseseq_amd import engine
            except Exception:  # (CPU-only runs: nothing to refresh)
                return
            engine.refresh_all_knobs()

    touched = []
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        orig_set(name, value, prepend)
        touched.append(name)
        refresh(name)

    def delenv(name, raising=True):
        orig_del(name, raising)
        touched.append(name)
        refresh(name)

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    switches = [n for n in touched if str(n).startswith(("DVS_", "HSA_CU_MASK", "ROC_GLOBAL_CU_MASK"))]
    if switches:  # (one re-read covers every switch the test touched, whichever it set first)
        refresh(switches[0])
