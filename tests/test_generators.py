"""The synthetic-network generators behind bench.py (host/generators.cpp): what a rank generates for its shard must be
exactly the full network's edges that start or end in the shard, every source's edges in the same order -- otherwise an N-GPU run would
simulate a different network than N = 1 without any error."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _sanafe_pkg  # noqa: E402

S = _sanafe_pkg.load()
pytestmark = pytest.mark.skipif(S.chip.lib() is None, reason="libsanafe_host not built")

N, DEG = 4096, 24


@pytest.mark.parametrize("window", [None, 1024, 4096])
def test_shards_are_the_slices_of_the_full_network(window):
    full = S.chip.generate_random_edges(N, DEG, seed=5, n_threads=3, window=window) if window else S.chip.generate_random_edges(N, DEG, seed=5, n_threads=3)
    fs, fd, fw = full
    assert len(fs) == N * DEG
    # every neuron has exactly DEG distinct targets, none of them itself; non-zero integer weights in +-8
    assert np.array_equal(np.bincount(fs, minlength=N), np.full(N, DEG))
    pairs = fs * N + fd
    assert len(np.unique(pairs)) == len(pairs)
    assert np.all(fw == np.round(fw)) and np.all(fw != 0) and np.all(np.abs(fw) <= 8)
    if window:
        dist = (fd - fs) % N
        dist = np.minimum(dist, N - dist)
        assert dist.max() <= window // 2  # targets come from the window centred on the neuron (ids wrap)
    for lo, hi in ((0, 1024), (1024, 2048), (3072, 4096), (1000, 1100)):
        ss, sd, sw = S.chip.generate_random_edges(N, DEG, seed=5, n_threads=2, shard=(lo, hi), window=window)
        keep = ((fs >= lo) & (fs < hi)) | ((fd >= lo) & (fd < hi))
        # the windowed generator lists the sources in ring order from lo - window/2 (…, N-1, 0, 1, …): what has to match is every source's
        # own sequence of edges (= the synapse order inside its axons); the order of the sources among themselves does
        # not enter the mapping (inbound axons are ordered by source core and neuron, src/chip.cpp:661-690)
        o, f = np.argsort(ss, kind="stable"), np.argsort(fs[keep], kind="stable")
        assert np.array_equal(ss[o], fs[keep][f]) and np.array_equal(sd[o], fd[keep][f]) and np.array_equal(sw[o], fw[keep][f]), (window, lo, hi)


def test_thread_count_does_not_change_the_network():
    a = S.chip.generate_random_edges(N, DEG, seed=9, n_threads=1, window=2048)
    b = S.chip.generate_random_edges(N, DEG, seed=9, n_threads=7, window=2048)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    c = S.chip.generate_random_edges(N, DEG, seed=10, n_threads=7, window=2048)
    assert not np.array_equal(a[1], c[1])
