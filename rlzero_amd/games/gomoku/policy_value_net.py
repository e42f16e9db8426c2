"""Policy-value network with the reference's architecture and parameter names
(rlzero/games/gomoku/policy_value_net.py:6-52): state_dicts are interchangeable.

  trunk   conv3x3(4->32) relu, conv3x3(32->64) relu, conv3x3(64->128) relu   (padding 1)
  policy  conv1x1(128->4) relu, fc(4S->S), log_softmax
  value   conv1x1(128->2) relu, fc(2S->64) relu, fc(64->1), tanh
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PolicyValueNet(nn.Module):

    def __init__(self, board_size: int, board_width: int = None, n_actions: int = None) -> None:
        """``PolicyValueNet(B)`` is the reference's square network.  ``board_width`` / ``n_actions``
        (extensions, e.g. Connect4: PolicyValueNet(6, 7, 7)) make the board board_size x board_width
        with n_actions policy outputs; the layers and their names are unchanged."""
        super().__init__()
        self.board_size = board_size
        self.board_width = board_width if board_width is not None else board_size
        cells = board_size * self.board_width
        self.n_actions = n_actions if n_actions is not None else cells
        # creation order = the reference's, so a seeded default init gives equal weights
        self.conv1 = nn.Conv2d(4, 32, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=3, padding=1)
        self.conv3 = nn.Conv2d(64, 128, kernel_size=3, padding=1)
        self.act_conv1 = nn.Conv2d(128, 4, kernel_size=1)
        self.act_fc1 = nn.Linear(4 * cells, self.n_actions)
        self.val_conv1 = nn.Conv2d(128, 2, kernel_size=1)
        self.val_fc1 = nn.Linear(2 * cells, 64)
        self.val_fc2 = nn.Linear(64, 1)

    def forward(self, obs: torch.Tensor):
        x = obs
        for conv in (self.conv1, self.conv2, self.conv3):
            x = F.relu(conv(x))
        batch = x.shape[0]
        act = F.relu(self.act_conv1(x)).reshape(batch, -1)
        log_probs = F.log_softmax(self.act_fc1(act), dim=1)
        val = F.relu(self.val_conv1(x)).reshape(batch, -1)
        val = F.relu(self.val_fc1(val))
        value = torch.tanh(self.val_fc2(val))
        return log_probs, value
