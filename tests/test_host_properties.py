"""Property tests (hypothesis) of the host-side helpers around the hot path."""
import numpy as np
from hypothesis import given, settings, strategies as st

import scpp_amd
from scpp_amd.distributed import pack_results, shard_range, unpack_results


@given(total=st.integers(0, 100000), world=st.integers(1, 16))
def test_shard_ranges_partition_the_batch(total, world):
    """Contiguous, disjoint, ordered shards that cover [0, total) and differ in size by at most one."""
    edges = [shard_range(total, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == total
    for (lo, hi), (lo2, _) in zip(edges, edges[1:]):
        assert lo <= hi == lo2
    sizes = [hi - lo for lo, hi in edges]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total


@settings(max_examples=25, deadline=None)
@given(B=st.integers(1, 6), K=st.integers(3, 12), seed=st.integers(0, 2**31 - 1))
def test_result_payload_round_trip(B, K, seed):
    rng = np.random.default_rng(seed)
    out = dict(X=rng.normal(size=(B, K, 14)), U=rng.normal(size=(B, K, 4)), sigma=rng.uniform(1, 20, size=B),
               nu_norm=rng.uniform(0, 1, size=B), sc_iters=rng.integers(0, 16, size=B).astype(np.int32),
               converged=rng.integers(0, 2, size=B).astype(np.int32))
    back = unpack_results(pack_results(out), K)
    for k in out:
        assert np.array_equal(back[k], out[k]), k


@settings(max_examples=50, deadline=None)
@given(K=st.integers(2, 20), t=st.floats(0.0, 30.0), total=st.floats(0.5, 30.0), foh=st.booleans(), seed=st.integers(0, 2**31 - 1))
def test_interpolated_input_stays_between_neighbouring_nodes(K, t, total, foh, seed):
    """commonFunctions.cpp:6-19: the result is a convex combination of two consecutive nodes (clamped to the last segment)."""
    rng = np.random.default_rng(seed)
    U = rng.normal(size=(1, K, 4))
    u = scpp_amd.interpolated_input(U, t, np.array([total]), foh)[0]
    dt = total / (K - 1)
    i = min(int(t / dt), K - 2)
    lo = np.minimum(U[0, i], U[0, i + 1] if foh else U[0, i])
    hi = np.maximum(U[0, i], U[0, i + 1] if foh else U[0, i])
    assert np.all(u >= lo - 1e-12) and np.all(u <= hi + 1e-12)


@given(seed=st.integers(0, 2**40), instance=st.integers(0, 2**30), draw=st.integers(0, 7))
def test_counter_based_uniform_is_in_range_and_stateless(seed, instance, draw):
    a = scpp_amd.counter_uniform(seed, instance, draw)
    assert -1.0 <= a < 1.0
    assert a == scpp_amd.counter_uniform(seed, instance, draw)
