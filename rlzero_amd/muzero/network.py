"""The learned model of MuZero (arXiv:1911.08265v2, section 3 and appendix F/G) for vector observations:
representation h(o) -> s, dynamics g(s, a) -> (r, s'), prediction f(s) -> (p, v); hidden states are
scaled to [0, 1] per sample (appendix G, "Training"), reward and value are scalar heads."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def scale_hidden(s):
    lo = s.min(dim=1, keepdim=True)[0]
    hi = s.max(dim=1, keepdim=True)[0]
    return (s - lo) / (hi - lo).clamp_min(1e-5)


class MuZeroNet(nn.Module):

    def __init__(self, obs_dim=4, n_actions=2, hidden=64):
        super().__init__()
        self.obs_dim, self.n_actions, self.hidden = obs_dim, n_actions, hidden
        self.rep1 = nn.Linear(obs_dim, hidden)
        self.rep2 = nn.Linear(hidden, hidden)
        self.dyn1 = nn.Linear(hidden + n_actions, hidden)
        self.dyn2 = nn.Linear(hidden, hidden)
        self.rew1 = nn.Linear(hidden, hidden)
        self.rew2 = nn.Linear(hidden, 1)
        self.pre1 = nn.Linear(hidden, hidden)
        self.pol = nn.Linear(hidden, n_actions)
        self.val = nn.Linear(hidden, 1)

    def representation(self, obs):
        return scale_hidden(self.rep2(F.relu(self.rep1(obs))))

    def dynamics(self, state, action):
        """state [b, hidden], action int64 [b] -> (next state, reward [b])."""
        x = torch.cat((state, F.one_hot(action, self.n_actions).to(state.dtype)), dim=1)
        h = F.relu(self.dyn1(x))
        nxt = scale_hidden(self.dyn2(h))
        reward = self.rew2(F.relu(self.rew1(h))).squeeze(1)
        return nxt, reward

    def prediction(self, state):
        """-> (policy logits [b, A], value [b])."""
        h = F.relu(self.pre1(state))
        return self.pol(h), self.val(h).squeeze(1)

    def initial_inference(self, obs):
        s = self.representation(obs)
        logits, value = self.prediction(s)
        return s, logits, value

    def recurrent_inference(self, state, action):
        nxt, reward = self.dynamics(state, action)
        logits, value = self.prediction(nxt)
        return nxt, reward, logits, value
