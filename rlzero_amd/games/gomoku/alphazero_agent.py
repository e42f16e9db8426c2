"""Evaluator + learner with the reference's interface
(rlzero/games/gomoku/alphazero_agent.py:12-125).

``policy_value_fn(env)`` is the per-position plug-in the reference's search calls once per
simulation.  ``AlphaZeroPlayer`` of this repository recognises a bound
``AlphaZeroAgent.policy_value_fn`` and instead runs ``policy_value_net`` once per
simulation STEP on the whole batch of leaves resident on the GPU (engine.NetEvaluator).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from .policy_value_net import PolicyValueNet


class AlphaZeroAgent(object):

    def __init__(self, board_size: int, learning_rate: float = 0.001, weight_decay: float = 1e-4,
                 device: str = 'cpu', board_width: int = None, n_actions: int = None) -> None:
        self.board_size = board_size
        self.board_width = board_width if board_width is not None else board_size
        self.device = device
        self.policy_value_net = PolicyValueNet(board_size, board_width, n_actions).to(device)
        self.optimizer = optim.Adam(self.policy_value_net.parameters(), lr=learning_rate,
                                    weight_decay=weight_decay)

    def _tensor(self, batch):
        return torch.as_tensor(np.array(batch), dtype=torch.float32).to(self.device)

    def policy_value_fn(self, game_env):
        """env -> (zip(legal actions, their probabilities), value): batch-1 forward;
        probabilities are exp(log_softmax) over ALL cells indexed by the legal ones (not
        renormalised), value is a Python float (alphazero_agent.py:31-46)."""
        legal = game_env.leagel_actions()
        obs = np.ascontiguousarray(game_env.current_state().reshape(-1, 4, self.board_size, self.board_width))
        log_probs, value = self.policy_value_net(torch.from_numpy(obs).float().to(self.device))
        probs = np.exp(log_probs.detach().cpu().numpy().flatten())
        return zip(legal, probs[legal]), value.item()

    def policy_value(self, state_batch):
        log_probs, value = self.policy_value_net(self._tensor(state_batch))
        return np.exp(log_probs.detach().cpu().numpy()), value.detach().cpu().numpy()

    def learn(self, state_batch, mcts_probs, target_vs):
        """One Adam step on (z - v)^2 - pi^T log p (L2 through weight_decay); returns
        (loss, policy entropy) as floats (alphazero_agent.py:59-86)."""
        self.policy_value_net.train()
        states, pis, zs = self._tensor(state_batch), self._tensor(mcts_probs), self._tensor(target_vs)
        log_probs, value = self.policy_value_net(states)
        loss = F.mse_loss(value.view(-1), zs) - torch.mean(torch.sum(pis * log_probs, dim=1))
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        entropy = -torch.mean(torch.sum(torch.exp(log_probs) * log_probs, dim=1))
        return loss.item(), entropy.item()

    def predict(self, state_batch):
        self.policy_value_net.eval()
        with torch.no_grad():
            log_probs, value = self.policy_value_net(torch.FloatTensor(state_batch).to(self.device))
        return np.exp(log_probs.cpu().numpy()), value.cpu().numpy()

    def save_model(self, save_dir: str, model_name: str = 'model.th', opt_name: str = 'optimizer.th'):
        """``save_dir`` is a DIRECTORY holding the two state_dicts (alphazero_agent.py:99-111)."""
        if not os.path.exists(save_dir):
            os.mkdir(save_dir)
        torch.save(self.policy_value_net.state_dict(), os.path.join(save_dir, model_name))
        torch.save(self.optimizer.state_dict(), os.path.join(save_dir, opt_name))
        print('save model successfully!')

    def restore(self, save_dir: str, model_name: str = 'model.th', opt_name: str = 'optimizer.th'):
        if not os.path.exists(save_dir):
            os.mkdir(save_dir)
        self.policy_value_net.load_state_dict(torch.load(os.path.join(save_dir, model_name)))
        self.optimizer.load_state_dict(torch.load(os.path.join(save_dir, opt_name)))
        print('restore model successfully!')
