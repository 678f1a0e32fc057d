"""-m gpu: vf_engine_opts.use_hip_graph (test or delete).  vf_engine_iterate replays its launch sequence
from a captured hipGraph; the capture is redone whenever something baked into it changes -- the number of trials, the
warm-start tail (cold solve / 1 appended keyframe / 2), the LM termination tolerances.  Required: the same bits as plain
launches over all of those changes, and evidence that replays really happened (no silent fallback)."""
import numpy as np
import pytest

from tests.test_gpu_ingest import _engine, _feed
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("form", ["partitioned", "one_wave_sweep", "assembling_sweep"])
def test_graph_replay_is_bit_identical_to_plain_launches(form):
    n, updates = 96, 10
    seqs = [synth.make_sequence(seed=500 + i, n_kf=n + updates + 1) for i in range(3)]
    opts = {} if form == "partitioned" else dict(chunks=1, sweep_two_sided_max=0, solve_assemble_min=1 if form == "assembling_sweep" else 0)
    eager = _engine(None, seqs, n, updates, **opts)
    graph = _engine(None, seqs, n, updates, use_hip_graph=True, **opts)
    calls = 0

    def both(fn):
        for e in (eager, graph):
            fn(e)

    def same(u):
        for w in range(3):
            np.testing.assert_array_equal(eager.get_states(w, u, n), graph.get_states(w, u, n))
            assert eager.read_lm(w) == graph.read_lm(w)

    both(lambda e: e.iterate(12)); calls += 1                 # cold solve
    same(0)
    both(lambda e: e.iterate(12)); calls += 1                 # same shape again: a replay without a capture
    same(0)
    k = n
    for u, trials in ((1, 5), (2, 5), (3, 3), (4, 5)):        # warm solves after one slide; the trial count changes
        both(lambda e: (e.ingest_tail(*_feed(seqs, k)), e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True), e.iterate(trials)))
        calls += 1
        k += 1
        same(u)
    both(lambda e: (e.ingest_tail(*_feed(seqs, k)), e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)))
    k += 1
    both(lambda e: (e.ingest_tail(*_feed(seqs, k)), e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True), e.iterate(5)))   # two slides, one solve
    calls += 1
    k += 1
    same(6)
    both(lambda e: e.set_convergence(1e-5, 1e-5))             # termination rule on: baked into the kernel arguments
    for u in (7, 8):
        both(lambda e: (e.ingest_tail(*_feed(seqs, k)), e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True), e.iterate(5)))
        calls += 1
        k += 1
        same(u)
    both(lambda e: e.set_convergence(0.0, 0.0))
    both(lambda e: (e.set_states(1, 20, e.get_states(1, 20, 1)), e.iterate(4)))   # a touched engine: cold again
    calls += 1
    same(8)
    enabled, captures, replays = graph.graph_info()
    assert enabled and replays == calls and 6 <= captures < calls, (enabled, captures, replays, calls)
    assert eager.graph_info() == (False, 0, 0)
    eager.close()
    graph.close()
