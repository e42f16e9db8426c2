"""CPU oracle for the AlphaZero self-play MCTS hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
``rlzero_amd`` never imports this package.

What it is: a from-scratch restatement, in plain Python, of the algorithm the reference implements in

* ``rlzero/mcts/node.py``            (tree node, UCT select, expand, backup)
* ``rlzero/mcts/alphazero_mcts.py``  (playout, simulate, player, tree reuse)
* ``rlzero/games/gomoku/gomoku_env.py`` (rules, win scan, observation planes)
* ``rlzero/games/gomoku/game.py``    (self-play / two-player game loop)
* ``rlzero/games/gomoku/policy_value_net.py`` + ``alphazero_agent.py`` (evaluator)

Pinning: the reference has NO tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, produced in the build container by ``tests/golden/gen_golden.py``
(which imports ``/root/reference``) and committed as fixtures under
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every fixture
bit-for-bit (visit counts, W as fp64 bit patterns, moves, winners, z,
observation planes) and pi to 1e-12.

Not of the reference: ``connect4_ref.py``, ``muzero_ref.py`` (build-defined rules / the published MuZero pseudocode: the
reference has neither -- parity unpinned by nature, as their headers say) and ``fp8_cross_ref.py`` (the float64 model of
the OPT-IN RZ_NET_SPLIT_F16_FP8 trunk arithmetic: it says what that mode should compute; ``tests/test_fp8_trunk.py``
says how far that is from the reference's forward in float64).
"""
