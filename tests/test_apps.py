"""The reference's app surface (diverse_seq/records.py:254-429, cluster.py:98-188, the `cogent3.app`
entry points of pyproject.toml:89-94) mirrored by diverseseq_amd.apps: constructor contract and
registration on the CPU, the reference's own app tests (tests/test_records.py:170-204,
tests/test_ctree.py:9-74) and oracle parity under the same seeded shuffle on the GPU."""
import importlib
import pathlib
import re

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, ROOT, read_fasta, str2arr


def test_entry_points_resolve_to_the_app_classes():
    text = (ROOT / "pyproject.toml").read_text()
    block = text.split('[project.entry-points."cogent3.app"]')[1].split("[", 1)[0]
    found = dict(re.findall(r'^(\w+)\s*=\s*"([\w.]+:\w+)"', block, flags=re.M))
    assert sorted(found) == ["dvs_ctree", "dvs_delta_jsd", "dvs_max", "dvs_nmost", "dvs_par_ctree"]
    for name, target in found.items():
        mod, attr = target.split(":")
        cls = getattr(importlib.import_module(mod), attr)
        assert cls.__name__ == name and callable(getattr(cls, "main"))


def test_constructor_contract():
    from diverseseq_amd import apps

    a = apps.dvs_max()  # records.py:258-267 defaults
    assert (a._min_size, a._max_size, a._stat, a._k, a._num_states) == (5, 30, "stdev", 6, 4)
    assert apps.dvs_max(include="Human")._include == ["Human"]
    assert apps.dvs_max(moltype="protein")._num_states == 20
    n = apps.dvs_nmost()  # records.py:328-335
    assert (n._n, n._k, n._include) == (10, 6, None)
    c = apps.dvs_ctree()  # cluster.py:102-111
    assert (c._k, c._sketch_size, c._distance_mode, c._mash_canonical) == (12, 3000, "mash", False)
    with pytest.raises(ValueError, match="Unexpected distance"):
        apps.dvs_ctree(distance_mode="blah")
    with pytest.raises(ValueError, match="Expected sketch size"):
        apps.dvs_ctree(sketch_size=None)
    with pytest.raises(ValueError, match="only supported for dna/rna"):
        apps.dvs_ctree(moltype="protein", mash_canonical_kmers=True)
    p = apps.dvs_par_ctree(max_workers=4, parallel=True, distance_mode="euclidean", sketch_size=None, k=5)
    assert p._max_workers == 4


@pytest.fixture(scope="module")
def brca1_text():
    raw = read_fasta(GOLDEN / "brca1.fasta")
    return {n: s.replace("-", "").replace("?", "") for n, s in raw.items()}


@pytest.mark.gpu
def test_select_apps_like_the_reference_tests(brca1_text):
    from diverseseq_amd import apps

    got = apps.dvs_max(k=1, min_size=2, max_size=5)(brca1_text)
    assert 2 <= len(got) <= 5
    got = apps.dvs_max(k=1, min_size=2, max_size=5, seed=123)(brca1_text)
    assert 2 <= len(got) <= 5
    for include in ("Human", ["Human"], ["Human", "Mouse"]):
        inc = {include} if isinstance(include, str) else set(include)
        assert inc <= set(apps.dvs_max(k=1, min_size=2, max_size=5, include=include)(brca1_text))
        assert inc <= set(apps.dvs_nmost(k=1, n=5, include=include)(brca1_text))
    assert len(apps.dvs_nmost(k=1, n=5)(brca1_text)) == 5
    # the same seeded shuffle as the reference (records.py:306-309, 366-369), then the oracle
    for seed, k, n in ((123, 4, 7), (5, 2, 10)):
        got = apps.dvs_nmost(k=k, n=n, seed=seed)(brca1_text)
        ids = list(brca1_text)
        np.random.default_rng(seed).shuffle(ids)
        exp = oracle.nmost([str2arr(brca1_text[i]) for i in ids], n, k, 4)
        assert set(got) == {ids[i] for i in exp.members()[0]}
        assert all(got[name] == brca1_text[name] for name in got)  # the caller's own sequences come back
    got = apps.dvs_max(k=3, min_size=5, max_size=20, stat="cov", seed=9)(brca1_text)
    ids = list(brca1_text)
    np.random.default_rng(9).shuffle(ids)
    exp = oracle.max_divergent([str2arr(brca1_text[i]) for i in ids], 5, 20, 3, 4, "cov")
    assert set(got) == {ids[i] for i in exp.members()[0]}


@pytest.mark.gpu
def test_delta_jsd_app(brca1_text):
    from diverseseq_amd import apps

    names = list(brca1_text)[:6]
    ref = {n: brca1_text[n] for n in names[:4]}
    app = apps.dvs_delta_jsd(ref, k=3)
    name, d = app((names[4], brca1_text[names[4]]))
    oset = oracle.SummedRecords.from_seqs([str2arr(s) for s in ref.values()], 3, 4)
    f, h = oracle.to_kfreqs(str2arr(brca1_text[names[4]]), 4, 3)
    assert name == names[4] and abs(d - oset.delta_jsd(f, h)) <= 1e-9
    assert app((names[0], ref[names[0]]))[1] == 0.0  # a member's own id (records.rs:71-73)
    assert np.isnan(app(("empty", ""))[1])            # records.py:424-425
    with pytest.raises(ValueError, match="zero-length sequences: bad"):
        apps.dvs_delta_jsd({"ok": "ACGT", "bad": ""}, k=1)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(k=3, sketch_size=400, distance_mode="mash"),
                                dict(k=3, sketch_size=None, distance_mode="euclidean")])
def test_ctree_apps_topology(brca1_text, kw):
    """reference tests/test_ctree.py:9-74"""
    from diverseseq_amd import apps
    from conftest import clades

    names = ["Human", "Chimpanzee", "Rhesus", "Horse"]
    seqs = {n: brca1_text[n] for n in names}
    expect = clades("(((Human,Chimpanzee),Rhesus),Horse);")
    for app in (apps.dvs_ctree(**kw), apps.dvs_par_ctree(max_workers=4, parallel=True, **kw)):
        assert clades(app(seqs)) == expect
