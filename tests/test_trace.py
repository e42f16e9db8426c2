"""rlzero_amd/trace.py: the device-side launch trace behind bench.py's `lane_timeline` (un-profiled evidence of the lanes' overlap)."""
import numpy as np
import pytest


def _records(launches):
    """launches: [(lane, kind, step, [(t0, t1, cu), ...workgroups])] -> the dict TraceBuffer.records() returns."""
    rows = [(t0, t1, kind, step, block, lane, cu) for lane, kind, step, wgs in launches for block, (t0, t1, cu) in enumerate(wgs)]
    cols = list(zip(*rows))
    return {k: np.array(c, dtype=np.int64) for k, c in zip(('t0', 't1', 'kind', 'step', 'block', 'lane', 'cu'), cols)}


def test_summary_of_a_known_schedule():
    """Two lanes of two games on a two-CU chip, every trunk workgroup 1000 ticks (10 us), tree steps of 400 ticks; lane 1 runs half a
    cycle behind lane 0: one trunk launch on the chip at any time, each CU under a trunk workgroup half of the time."""
    from rlzero_amd.trace import KIND_TREE, KIND_TRUNK, summarise
    launches = []
    for lane, phase in ((0, 0), (1, 1000)):
        for step in range(50):
            t = phase + 2000 * step
            launches.append((lane, KIND_TRUNK, step, [(t, t + 1000, 0), (t, t + 1000, 1)]))
            launches.append((lane, KIND_TREE, step, [(t + 1000, t + 1400, 0), (t + 1000, t + 1400, 1)]))
    out = summarise(_records(launches), n_cus=2)
    assert out['trunk_launches_traced'] == 100 and out['lanes']['0']['trunk_launches'] == 50
    assert abs(out['launches_in_flight'] - 1.0) < 1e-6 and out['trunk_launches_on_chip']['with_1'] > 0.999
    assert abs(out['cu_time_in_trunk'] - 1.0) < 0.01          # (two workgroups of 10 us per 10 us on two CUs)
    assert out['lanes']['1']['step_cycle_us'] == 20.0 and out['lanes']['0']['trunk_launch_us'] == 10.0
    assert out['lanes']['0']['tree_launch_us'] == 4.0 and out['trunk_workgroup_us']['mean'] == 10.0
    assert abs(out['sims_per_sec_in_window'] - 2e5) / 2e5 < 0.03   # 2 boards per 10 us


def test_coverage_counts_overlaps():
    from rlzero_amd.trace import _coverage
    cover, depth = _coverage([(0, 10), (5, 15), (5, 20), (6, 8)], 0, 20)
    assert np.allclose(cover, [0.0, 0.5, 0.25, 0.25])   # depth 1 on [0,5) + [15,20), 2 on [10,15), 3 or 4 on [5,10)
    assert abs(depth - (10 + 10 + 15 + 2) / 20.0) < 1e-12


@pytest.mark.gpu
def test_trace_of_the_four_lane_layout():
    """The shipped layout (512 games, four lanes, hipGraphs, pipelined moves) with the trace attached, 160 simulations per move:
    every lane's last search is there in full -- 160 trunk launches of 128 workgroups and 150 tree steps --, the CUs are mostly
    under trunk workgroups, and more than one lane's trunk is on the chip at a time."""
    import torch
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.trace import measure
    torch.manual_seed(0)
    net = PolicyValueNet(15).to('cuda:0').eval()
    # (512 games are ONE resident lane by default since round 6 -- k_delta_res, two games per CU; the trace reads the lanes of the
    # two-launch step, the layout of more than 2 x CUs games)
    out = measure(net, 15, 5, n_games=512, n_playout=160, warm_moves=2, lanes=4, resident_search=False)
    # (the last step of a graph chunk of 16 is backup only -- the next chunk opens with its own selection -- and is not traced: 150 tree steps)
    assert out['lanes_in_layout'] == 4 and out['records'] == 4 * 128 * (160 + 150)
    for ln in '0123':
        lane = out['lanes'][ln]
        assert lane['trunk_launches'] == 160 and lane['workgroups_per_trunk_launch'] == 128.0
        assert 5.0 < lane['tree_launch_us'] < 60.0 and 10.0 < lane['trunk_launch_us'] < 120.0
    # (searches this short overlap only partly -- every lane's host step is a fifth of its search: the full-size figures are
    # on the bench line -- so the bounds are loose)
    assert 0.3 < out['cu_time_in_trunk'] <= 1.0 and out['launches_in_flight'] > 1.0 and 200 <= out['cus_seen'] <= 256
    assert out['window'] in ('all lanes searching', "union of the lanes' searches")   # (the second when these short searches barely overlap)
    assert 8.0 < out['trunk_workgroup_us']['mean'] < 40.0 and out['sims_per_sec_in_window'] > 3e6
