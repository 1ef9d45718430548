"""The multifrontal solve is one fixed arithmetic laid over several streams (look-ahead across tree levels, side streams, event
schemes chosen by regime). Scheduling must not change a bit of the result: a solve with the look-ahead across levels switched off
(every level strictly after the one below, on one stream) must reproduce the default solve EXACTLY — a missing dependency
between streams shows up here as a difference, long before it shows up as a wrong answer. The two kernel-form switches change the
summation order and are held to rounding level instead."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HELPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "solve_once.py")


def _solve(tmp_path, tag, name, **env):
    out = str(tmp_path / f"{tag}.npz")
    r = subprocess.run([sys.executable, HELPER, name, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("name", ["mh01", "mh123"])
def test_look_ahead_across_levels_changes_no_bit(tmp_path, name):
    a = _solve(tmp_path, "default", name)
    b = _solve(tmp_path, "default_again", name)
    c = _solve(tmp_path, "no_lookahead", name, COVGPU_ND_LOOKAHEAD="0")
    for k in ("pose", "sb", "lm", "cost"):
        assert np.array_equal(a[k], b[k]), f"two default solves differ in {k}: the solve is not deterministic"
        assert np.array_equal(a[k], c[k]), f"look-ahead on / off differ in {k}: a dependency between streams is missing"
    assert np.array_equal(a["acc"], c["acc"])
    # round 4: the event scheme of a multi-panel front (an event behind every kernel | round 3's single event per panel) and the border
    # tiles the first trailing update starts from zero instead of having them cleared — both leave the arithmetic alone
    d = _solve(tmp_path, "single_event_cleared", name, COVGPU_CHAIN_PAIRS="700", COVGPU_BETA0="0")
    for k in ("pose", "sb", "lm", "cost"):
        assert np.array_equal(a[k], d[k]), f"event scheme / un-cleared border tiles differ in {k}"


@pytest.mark.parametrize("name", ["mh01", "mh12345"])
def test_device_flag_ordering_equals_event_ordering(tmp_path, name):
    """Round 6: the streams of a context are ordered through device flags (a signal kernel behind the producer, a polling gate in front of the
    consumer: CholAux::record / wait) instead of HIP events. The arithmetic is untouched: both orderings give the same bits; a dependency the
    flags miss shows up here."""
    a = _solve(tmp_path, "gates", name)
    b = _solve(tmp_path, "events", name, COVGPU_GATES="0")
    for k in ("pose", "sb", "lm", "cost", "acc"):
        assert np.array_equal(a[k], b[k]), f"device-flag ordering and event ordering differ in {k}"


@pytest.mark.parametrize("name", ["mh01", "mh123"])
def test_extend_add_from_records_changes_no_bit(tmp_path, name):
    """Round 6: the extend-add reads one record per (tile, contributing child) — header, child and row maps behind one address — instead of chasing
    tile -> child entries -> row maps (k_nd_extend_rec | k_nd_extend, COVGPU_EXT_RECORDS=0). Children that reach neither the tile's rows nor its columns
    are skipped; their terms were zero. Same sums in the same order: the same bits."""
    a = _solve(tmp_path, "records", name)
    b = _solve(tmp_path, "chased", name, COVGPU_EXT_RECORDS="0")
    for k in ("pose", "sb", "lm", "cost", "acc"):
        assert np.array_equal(a[k], b[k]), f"the two extend-add kernels differ in {k}"


def test_no_solve_falls_back_to_events():
    """Device-flag ordering must hold on every path without ever reaching its timeout: GBA and — the case that hung in round 6, when two streams
    stored into one flag slot (one panel event recorded on the chain's stream at one level and on a side stream at the next) — the pose graph,
    on one context, one after the other (covgpu_get_layout: stream_ordering 1 = flags, -1 = fell back to events)."""
    from covins_amd import backend, mapdata, synth
    m = synth.make_map(synth.config_named("mh123"))
    p, _ = mapdata.flatten_gba(m, visual_only=False, loop_loss=True)
    q = mapdata.flatten_pgo(m, {}, mapdata.PgoParams())[0]
    ctx = backend.Context(0)
    try:
        for _ in range(2):
            ctx.gba_solve(p, backend.default_options(max_iterations=4))
            assert ctx.layout()["stream_ordering"] == 1
            sol, res = ctx.pgo_solve(q, backend.default_options(max_iterations=10))
            assert ctx.layout()["stream_ordering"] == 1 and res.final_cost < res.initial_cost
    finally:
        ctx.close()


def test_gate_timeout_falls_back_to_events(tmp_path):
    """A gate that cannot be satisfied in time raises the context's give-up flag; the solve is repeated with HIP events — and without the
    hand-overs inside a launch (k_bwd_pipe), which rest on the same kind of wait — and the result is that form's (here the timeout is set below
    a kernel's duration: 1e-7 s)."""
    a = _solve(tmp_path, "events", "mh01", COVGPU_GATES="0", COVGPU_BWD_PIPE="0")
    b = _solve(tmp_path, "gates_timing_out", "mh01", COVGPU_GATE_TIMEOUT_S="0.0000001", COVGPU_GATE_TIMEOUT_MIN="0")
    for k in ("pose", "sb", "lm", "cost", "acc"):
        assert np.array_equal(a[k], b[k])
    assert b["ordering"] == -1


def test_backward_pipeline_timeout_falls_back_to_a_launch_per_tile(tmp_path):
    """The pipelined backward substitution (one workgroup per interior tile, the solved tiles handed over inside the launch) gives up the same way:
    a hand-over that does not arrive in time raises the give-up flag, the solve is repeated with a launch per tile. Here the stream ordering is
    already HIP events (so only the pipeline can time out), the first tile of every chain withholds its result (COVGPU_PIPE_FAULT: a workgroup that
    keeps up never has to wait otherwise), the clock is looked at after every poll and the limit is 1e-7 s."""
    a = _solve(tmp_path, "events", "mh01", COVGPU_GATES="0", COVGPU_BWD_PIPE="0")
    b = _solve(tmp_path, "pipe_timing_out", "mh01", COVGPU_GATES="0", COVGPU_GATE_TIMEOUT_S="0.0000001", COVGPU_GATE_TIMEOUT_MIN="0", COVGPU_PIPE_SPIN_CHECK="1",
               COVGPU_PIPE_FAULT="1")
    for k in ("pose", "sb", "lm", "cost", "acc"):
        assert np.array_equal(a[k], b[k])
    assert b["ordering"] == -1 and a["ordering"] == 0


def test_kernel_form_switches_stay_at_rounding_level(tmp_path):
    a = _solve(tmp_path, "default", "mh01")
    # (round_3_tail: the trust-region tail as ~25 launches with the second J*v pass on the combined step, instead of k_tail.hip's
    #  one pass over both dogleg directions: the model decrease is the same quadratic form, summed in another order)
    for tag, env in (("full_tiles", dict(COVGPU_QUARTER_MAX="0")), ("per_tile_backward", dict(COVGPU_ND_BWD_FUSED="0")), ("no_backward_pipeline", dict(COVGPU_BWD_PIPE="0")), ("bottom_levels_one_by_one", dict(COVGPU_BWD_TREE="0")),
                     ("round_3_tail", dict(COVGPU_TAIL="0"))):
        b = _solve(tmp_path, tag, "mh01", **env)
        assert np.array_equal(a["acc"], b["acc"])
        assert np.allclose(a["cost"], b["cost"], rtol=1e-8)
        assert np.abs(a["pose"] - b["pose"]).max() < 1e-8 and np.abs(a["sb"] - b["sb"]).max() < 1e-8
