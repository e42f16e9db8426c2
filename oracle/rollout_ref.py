"""Oracle: the pure-MCTS evaluation opponent (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates rlzero/mcts/rollout_mcts.py: ``_playout`` :23-47, ``_evaluate`` :49-74 (including its
perspective rule: the value is taken for the player to move AFTER the rollout), ``simulate``
:76-81 (most visited child, first maximum), ``rollout_policy`` :96-100, ``policy_value_fn``
:102-108, ``RolloutPlayer.get_action`` :129-136 (tree reset after every move).

``rand(k)`` supplies the k numbers whose arg-max picks the rollout move (numpy's global
``np.random.rand`` in the reference); tests inject recorded or synthetic streams.
"""
import numpy as np

from .mcts_ref import RefNode, backup, expand, select_child


class RefRolloutSearch(object):

    def __init__(self, n_playout=1000, c_puct=5.0, n_limit=1000, rand=None):
        self.root = RefNode(None, 1.0)
        self.n_playout = n_playout
        self.c_puct = c_puct
        self.n_limit = n_limit
        self.rand = rand if rand is not None else np.random.rand
        self.sim_index = 0

    def evaluate(self, env):
        winner = -1
        ply = 0
        for ply in range(self.n_limit):
            ended, winner = env.game_end_winner()
            if ended:
                break
            legal = env.leagel_actions()
            scores = self.rand(len(legal))
            best = 0
            for i in range(1, len(legal)):  # max(..., key=itemgetter(1)): first maximum
                if scores[i] > scores[best]:
                    best = i
            env.step(legal[best])
        else:
            print('WARNING: rollout reached move limit')
        if winner == -1:
            return 0
        return 1.0 if winner == env.current_player() else -1.0

    def playout(self, env):
        node = self.root
        while node.kids:
            action, node = select_child(node, self.c_puct)
            env.step(action)
        legal = env.leagel_actions()
        ended, _ = env.game_end_winner()
        if not ended:
            expand(node, [(a, 1.0 / len(legal)) for a in legal])
        leaf_value = self.evaluate(env)
        backup(node, -leaf_value)
        self.sim_index += 1

    def simulate(self, env):
        for _ in range(self.n_playout):
            self.playout(env.clone())
        best = 0
        for i, kid in enumerate(self.root.kids):
            if kid.n > self.root.kids[best].n:
                best = i
        return self.root.acts[best]

    def update_with_move(self, last_move):
        if last_move in self.root.acts:
            self.root = self.root.child(last_move)
            self.root.parent = None
        else:
            self.root = RefNode(None, 1.0)


class RefRolloutPlayer(object):

    def __init__(self, n_playout=1000, c_puct=5, rand=None):
        self.mcts = RefRolloutSearch(n_playout, c_puct, rand=rand)
        self.player_id = 0

    def set_player_id(self, player_id):
        self.player_id = player_id

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def get_action(self, env, **kwargs):
        if not env.leagel_actions():
            print('WARNING: the board is full')
            return None
        move = self.mcts.simulate(env)
        self.mcts.update_with_move(-1)
        return move
