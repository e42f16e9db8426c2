#!/usr/bin/env python3
"""profiles/r04/lane_timeline.json: the schedule of the shipped layout from the device-side launch trace (rlzero_amd/trace.py) --
no profiler in the way.  `python profiles/lane_timeline.py [--games 512] [--lanes N] [--moves 2] > gpurun_out/lane_timeline.json`"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--games', type=int, default=512)
    ap.add_argument('--lanes', type=int, default=0)
    ap.add_argument('--moves', type=int, default=3, help='pipelined moves played under the trace (the last search of every lane is what is read)')
    ap.add_argument('--playouts', type=int, default=800)
    args = ap.parse_args()
    import torch
    import rlzero_amd  # noqa: F401  (claims the hardware queues before the runtime starts)
    from rlzero_amd.games.gomoku.policy_value_net import PolicyValueNet
    from rlzero_amd.trace import measure
    torch.manual_seed(0)
    net = PolicyValueNet(15).to('cuda:0').eval()
    kw = {'lanes': args.lanes} if args.lanes > 0 else {}
    out = measure(net, 15, 5, n_games=args.games, n_playout=args.playouts, warm_moves=args.moves, **kw)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
