"""`dvs ctree` path: distances on the GPU, average-linkage tree on the host.

Mirrors diverse_seq/cluster.py: `make_cluster_tree` (:191-237 -- sklearn
AgglomerativeClustering(metric="precomputed", linkage="average"), children_ folded into a
nested tuple, printed without quotes) and the argument checks of `dvs_ctree.__init__`
(:113-162).  The reference hands the string to cogent3's make_tree; here it is returned as a
Newick string (clustering is O(N^2..N^3) on a tiny matrix and stays on
the host, as in the reference).
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np

from . import distance


def nested_tuple_tree(seq_names: Sequence[str], pairwise_distances: np.ndarray):
    """diverse_seq/cluster.py:216-230"""
    from sklearn.cluster import AgglomerativeClustering

    clustering = AgglomerativeClustering(metric="precomputed", linkage="average")
    clustering.fit(np.asarray(pairwise_distances, dtype=np.float64))
    tree = {i: seq_names[i] for i in range(len(seq_names))}
    node = len(seq_names)
    for left, right in clustering.children_:
        tree[node] = (tree.pop(int(left)), tree.pop(int(right)))
        node += 1
    return tree[node - 1]


def make_cluster_tree(seq_names: Sequence[str], pairwise_distances: np.ndarray) -> str:
    """-> Newick string of the average-linkage tree (diverse_seq/cluster.py:231-233)"""
    if len(seq_names) < 2:
        raise ValueError("need at least two sequences to build a tree")
    return str(nested_tuple_tree(seq_names, pairwise_distances)).replace("'", "") + ";"


def ctree(seqs: dict, *, k: int = 12, sketch_size: int | None = 3000, distance_mode: str = "mash",
          mash_canonical_kmers: bool | None = None, num_states: int = 4) -> str:
    """sequences {name: uint8 codes} -> Newick string (dvs_ctree.main, cluster.py:164-188).
    Argument checks as dvs_ctree.__init__ (cluster.py:139-162)."""
    if mash_canonical_kmers is None:
        mash_canonical_kmers = False
    if distance_mode not in ("mash", "euclidean"):
        raise ValueError(f"Unexpected distance {distance_mode!r}.")
    if distance_mode == "mash" and sketch_size is None:
        raise ValueError("Expected sketch size for mash distance measure.")
    if distance_mode != "mash" and sketch_size is not None:
        raise ValueError("Sketch size should only be specified for the mash distance.")
    if distance_mode != "mash" and mash_canonical_kmers:
        raise ValueError("Canonical kmers should only be specified for the mash distance.")
    names = list(seqs)
    arrays = [seqs[n] for n in names]
    if distance_mode == "mash":
        dists = distance.mash_distances(arrays, k, int(sketch_size), num_states, mash_canonical_kmers)
    else:
        dists = distance.euclidean_distances(arrays, k, num_states)
    return make_cluster_tree(names, dists)
