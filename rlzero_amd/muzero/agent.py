"""MuZero learner: K-step unrolled training of the learned model (arXiv:1911.08265v2, appendix G and the
pseudocode's ``update_weights``): value and reward by squared error (scalar heads), policy by cross-entropy
against the search visit distribution, gradient of the unrolled steps scaled by 1/K, the gradient flowing into
the dynamics function halved at every step."""
import torch
import torch.nn.functional as F

from .network import MuZeroNet


def scale_gradient(x, scale):
    return x * scale + x.detach() * (1.0 - scale)


class MuZeroAgent(object):

    def __init__(self, obs_dim=4, n_actions=2, hidden=64, lr=3e-3, weight_decay=1e-4, device='cuda:0'):
        self.device = torch.device(device)
        self.n_actions = n_actions
        self.net = MuZeroNet(obs_dim, n_actions, hidden).to(self.device)
        self.optimizer = torch.optim.Adam(self.net.parameters(), lr=lr, weight_decay=weight_decay)

    def learn(self, batch):
        """batch = ReplayBuffer.sample(...) -> (loss, value loss, reward loss, policy loss) floats."""
        obs, actions, tv, tr, tp, mask = (torch.from_numpy(a).to(self.device) for a in batch)
        K = actions.shape[1]
        net = self.net
        net.train()
        state, logits, value = net.initial_inference(obs)
        lv = F.mse_loss(value, tv[:, 0], reduction='none')
        lr_ = torch.zeros_like(lv)
        lp = -(tp[:, 0] * F.log_softmax(logits, dim=1)).sum(dim=1) * mask[:, 0]
        for k in range(1, K + 1):
            state, reward, logits, value = net.recurrent_inference(state, actions[:, k - 1])
            g = 1.0 / K
            lv = lv + scale_gradient(F.mse_loss(value, tv[:, k], reduction='none'), g)
            lr_ = lr_ + scale_gradient(F.mse_loss(reward, tr[:, k], reduction='none'), g)
            lp = lp + scale_gradient(-(tp[:, k] * F.log_softmax(logits, dim=1)).sum(dim=1) * mask[:, k], g)
            state = scale_gradient(state, 0.5)
        loss = (0.25 * lv + lr_ + lp).mean()
        self.optimizer.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 5.0)
        self.optimizer.step()
        net.eval()
        return float(loss.item()), float(lv.mean().item()), float(lr_.mean().item()), float(lp.mean().item())

    def save_model(self, path):
        torch.save({'model': self.net.state_dict(), 'optimizer': self.optimizer.state_dict()}, path)

    def load_model(self, path):
        ck = torch.load(path, map_location=self.device)
        self.net.load_state_dict(ck['model'])
        self.optimizer.load_state_dict(ck['optimizer'])
