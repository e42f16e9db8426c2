"""Un-profiled evidence of what the lanes' kernels do on the chip: the device-side launch trace (csrc/rz_trace.h).

With a trace buffer attached (``rz_trace_attach`` / ``rz_net_trace_attach``, before the hipGraphs are captured) every trunk
workgroup and every tree-step workgroup of the deferred-priors route leaves {start, end (100 MHz constant clock), lane tag, step,
block, CU} in its lane's buffer (four plain stores, no atomic; a search overwrites the one before it).  rocprofv3 serialises the four hardware queues of the shipped layout (512 games: 3.2 M instead of 10 M simulations / s
under it), so its per-dispatch durations describe kernels ALONE on the chip; this trace runs inside the graphs at nearly full
speed (``measure`` reports the traced run's own rate beside the figures).

``measure`` plays a few moves of the BASELINE layout with the trace on and reduces the records to the schedule's figures:
per-lane launch intervals, how much of the time 0 / 1 / 2 / 3+ trunk launches are on the chip, how busy the CUs are, and the
``launches_in_flight`` that bench.py used to estimate from a ratio of averages.
"""
import time

import numpy as np

KIND_TRUNK, KIND_TREE = 1, 2
TICK_US = 0.01   # s_memrealtime: 100 MHz


class TraceBuffer(object):
    """The trace buffer of ONE lane: 2 kinds x ``slots`` steps x ``blocks`` games records, addressed by (kind, step, block) -- no
    atomics on the device, and a search overwrites the one before it: what is read is the lane's LAST search."""

    def __init__(self, torch, device, slots, blocks):
        self.torch, self.slots, self.blocks = torch, int(slots), int(blocks)
        self.buf = torch.zeros(2 + 4 * 2 * self.slots * self.blocks, dtype=torch.int64, device=device)
        self.buf[0] = self.slots
        self.buf[1] = self.blocks

    def ptr(self):
        import ctypes
        return ctypes.c_void_p(self.buf.data_ptr())

    def reset(self):
        self.buf[2:] = 0

    def records(self, lane):
        """-> dict of numpy arrays over the records present: t0, t1 (ticks), kind, step, block, lane, cu (a key unique per CU)."""
        host = self.buf.cpu().numpy().view(np.uint64)
        r = host[2:].reshape(-1, 4)
        r = r[r[:, 1] != 0]
        meta, hw = r[:, 2], r[:, 3]
        xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
        # HW_REG_HW_ID (gfx9): wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13] (+ more SE bits on wider parts)
        cu_in_xcc = ((hw >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64)   # CU, SH, SE bits together
        return {'t0': r[:, 0].astype(np.int64), 't1': r[:, 1].astype(np.int64),
                'kind': (meta >> np.uint64(56)).astype(np.int64), 'step': ((meta >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(np.int64),
                'block': (meta & np.uint64(0xFFFFFFFF)).astype(np.int64), 'lane': np.full(len(r), lane, dtype=np.int64),
                'cu': xcc * 512 + cu_in_xcc}


def _coverage(intervals, t_lo, t_hi):
    """intervals [(a, b)] -> {k: share of [t_lo, t_hi) covered by exactly k of them (3 = 3 or more)}, mean count."""
    ev = []
    for a, b in intervals:
        a, b = max(a, t_lo), min(b, t_hi)
        if b > a:
            ev.append((a, 1))
            ev.append((b, -1))
    ev.sort()
    cover, depth, at, area = [0.0, 0.0, 0.0, 0.0], 0, t_lo, 0.0
    for t, d in ev:
        cover[min(depth, 3)] += t - at
        area += depth * (t - at)
        at, depth = t, depth + d
    cover[min(depth, 3)] += t_hi - at
    span = float(t_hi - t_lo)
    return [c / span for c in cover], area / span


def summarise(rec, n_cus=256):
    """The schedule's figures from the records of the lanes' last searches (see the module docstring)."""
    out = {'records': int(len(rec['t0']))}
    if not len(rec['t0']):
        return out
    trunk = rec['kind'] == KIND_TRUNK
    tree = rec['kind'] == KIND_TREE
    lane = rec['lane']
    # the window: the time during which EVERY lane's last search is running (the lanes' searches start and end a host step apart),
    # trimmed by 2 % at both ends
    t_lo = max(int(rec['t0'][trunk & (lane == ln)].min()) for ln in set(lane.tolist()))
    t_hi = min(int(rec['t1'][trunk & (lane == ln)].max()) for ln in set(lane.tolist()))
    out['window'] = 'all lanes searching'
    if t_hi - t_lo < 100:   # (searches too short to overlap: the union of the lanes' searches instead)
        t_lo, t_hi = int(rec['t0'][trunk].min()), int(rec['t1'][trunk].max())
        out['window'] = 'union of the lanes\' searches'
    trim = (t_hi - t_lo) // 50
    t_lo, t_hi = t_lo + trim, t_hi - trim
    span_us = (t_hi - t_lo) * TICK_US
    out['window_us'] = round(span_us, 1)
    # a lane's launches of one kind follow each other in time, and the workgroups of a launch share its step number (the slot of
    # the deferred store): a new launch = the step changes along the lane's records in start order
    def launches_of(kind):
        per = {}
        for ln in sorted(set(lane.tolist())):
            sel = (lane == ln) & (rec['kind'] == kind)
            order = np.argsort(rec['t0'][sel])
            t0s, t1s, steps = rec['t0'][sel][order], rec['t1'][sel][order], rec['step'][sel][order]
            groups, cur_step, cur = [], None, None
            for a, b, s_ in zip(t0s, t1s, steps):
                # workgroups of one launch share the step and start within its span; a new launch = the step changes
                if cur is None or s_ != cur_step:
                    if cur is not None:
                        groups.append(cur)
                    cur_step, cur = s_, [int(a), int(b), 1]
                else:
                    cur[0], cur[1], cur[2] = min(cur[0], int(a)), max(cur[1], int(b)), cur[2] + 1
            if cur is not None:
                groups.append(cur)
            per[ln] = groups
        return per
    trunk_l, tree_l = launches_of(KIND_TRUNK), launches_of(KIND_TREE)
    all_trunk = [(a, b) for g in trunk_l.values() for a, b, _ in g]
    cover, depth = _coverage(all_trunk, t_lo, t_hi)
    out['trunk_launches_on_chip'] = {'share_of_time_with_0': round(cover[0], 4), 'with_1': round(cover[1], 4), 'with_2': round(cover[2], 4),
                                     'with_3_or_more': round(cover[3], 4), 'mean': round(depth, 3)}
    # CUs: the share of (CU x time) with AT LEAST ONE trunk workgroup on the CU (a k_trunk_rows workgroup holds its CU's LDS alone; two
    # receptive-field workgroups, k_trunk_delta, fit side by side), and the mean number of trunk workgroups on a CU
    wg_busy = float(np.clip(np.minimum(rec['t1'][trunk], t_hi) - np.maximum(rec['t0'][trunk], t_lo), 0, None).sum())
    cus_seen = int(len(set(rec['cu'][trunk].tolist())))
    out['cus_seen'] = cus_seen
    covered = 0.0
    t0s, t1s, cus = rec['t0'][trunk], rec['t1'][trunk], rec['cu'][trunk]
    order = np.argsort(cus, kind='stable')
    bounds = np.flatnonzero(np.diff(cus[order])) + 1
    for idx in np.split(order, bounds):
        cover, _ = _coverage(list(zip(t0s[idx].tolist(), t1s[idx].tolist())), t_lo, t_hi)
        covered += 1.0 - cover[0]
    out['cu_time_in_trunk'] = round(covered / float(n_cus), 4)
    out['cu_idle_of_trunk'] = round(1.0 - out['cu_time_in_trunk'], 4)
    out['trunk_workgroups_per_cu'] = round(wg_busy / ((t_hi - t_lo) * float(n_cus)), 4)
    out['trunk_workgroup_us'] = {'mean': round(float((rec['t1'][trunk] - rec['t0'][trunk]).mean()) * TICK_US, 2),
                                 'p10': round(float(np.percentile(rec['t1'][trunk] - rec['t0'][trunk], 10)) * TICK_US, 2),
                                 'p90': round(float(np.percentile(rec['t1'][trunk] - rec['t0'][trunk], 90)) * TICK_US, 2)}
    out['tree_workgroup_us'] = {'mean': round(float((rec['t1'][tree] - rec['t0'][tree]).mean()) * TICK_US, 2),
                                'p90': round(float(np.percentile(rec['t1'][tree] - rec['t0'][tree], 90)) * TICK_US, 2)} if tree.any() else None
    lanes = {}
    for ln in sorted(trunk_l):
        tl, rl = trunk_l[ln], tree_l.get(ln, [])
        dur = np.array([b - a for a, b, _ in tl], dtype=np.float64) * TICK_US
        starts = np.array([a for a, _, _ in tl], dtype=np.float64) * TICK_US
        cycle = np.diff(starts)
        cycle = cycle[cycle < 10 * np.median(cycle)] if len(cycle) else cycle   # (the host step between two moves is not a cycle)
        rdur = np.array([b - a for a, b, _ in rl], dtype=np.float64) * TICK_US
        lanes[str(ln)] = {'trunk_launches': len(tl), 'trunk_launch_us': round(float(dur.mean()), 2),
                          'tree_launch_us': round(float(rdur.mean()), 2) if len(rdur) else None,
                          'step_cycle_us': round(float(np.median(cycle)), 2) if len(cycle) else None,
                          'workgroups_per_trunk_launch': round(float(np.mean([c for _, _, c in tl])), 1)}
    out['lanes'] = lanes
    n_launch = sum(len(g) for g in trunk_l.values())
    mean_launch = float(np.mean([b - a for a, b in all_trunk])) * TICK_US
    out['trunk_launch_us_mean'] = round(mean_launch, 2)
    # launches in flight: the time launches spend on the chip / the wall time they share = the coverage depth over busy + idle time
    out['launches_in_flight'] = round(depth, 3)
    inside = trunk & (rec['t0'] >= t_lo) & (rec['t0'] < t_hi)
    out['sims_per_sec_in_window'] = round(float(inside.sum()) / span_us * 1e6, 1)   # one trunk workgroup = one leaf = one simulation
    out['trunk_launches_traced'] = int(n_launch)
    return out


def measure(net_module, board=15, n_in_row=5, n_games=512, n_playout=800, warm_moves=3, device='cuda:0', seed=0, device_moves=True, **kw):
    """Play ``warm_moves`` + 1 pipelined moves of the shipped layout with a trace buffer attached to every lane; -> summary of the
    lanes' LAST searches plus 'sims_per_sec_traced' (the whole run under the trace, host steps included) for comparison with the
    untraced bench."""
    import torch
    from .selfplay import BatchedSelfPlay
    traces = []

    def attach(sp):
        for lane in sp.lanes:
            tb = TraceBuffer(torch, device, n_playout, lane.eng.n_games)
            traces.append(tb)
            lane.eng.lib.rz_trace_attach(lane.eng.handle, tb.ptr())
            lane.evaluator.hip.lib.rz_net_trace_attach(lane.evaluator.hip.handle, tb.ptr())

    sp = BatchedSelfPlay.for_network(net_module, board, n_in_row, n_games=n_games, n_playout=n_playout, device=device, seed=seed,
                                     before_warm=attach, **kw)
    try:
        if device_moves:   # the move step on the device (BatchedSelfPlay.play_move_device): what bench.py times by default
            sp.device_attach(queue_capacity=4 * n_games)
            sp.device_queue(range(4 * n_games))
            step = sp.play_move_device
        else:
            sp._start(range(n_games), range(n_games))
            sp._set_active()
            step = sp.play_move_pipelined
        step()
        torch.cuda.synchronize()
        if device_moves:
            sp.device_drain()
        for tb in traces:
            tb.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(warm_moves):
            step()
        torch.cuda.synchronize()
        if device_moves:
            sp.device_drain()
        dt = time.perf_counter() - t0
        parts = [tb.records(i) for i, tb in enumerate(traces)]
        rec = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        out = summarise(rec, n_cus=torch.cuda.get_device_properties(torch.device(device)).multi_processor_count)
        out['lanes_in_layout'] = len(sp.lanes)
        out['move_step'] = 'device' if device_moves else 'host, pipelined'
        out['games'] = n_games
        # (each call enqueues the next move's search before it returns; the last search ends at the synchronize)
        out['sims_per_sec_traced'] = round(warm_moves * n_games * n_playout / dt, 1)
        return out
    finally:
        for lane in sp.lanes:
            lane.eng.lib.rz_trace_attach(lane.eng.handle, None)
            lane.evaluator.hip.lib.rz_net_trace_attach(lane.evaluator.hip.handle, None)
            lane.eng.close()
