"""Small deterministic stand-in network used for fixtures (oracle; test infrastructure only)."""
import torch
import torch.nn as nn


class ToyEps(nn.Module):
    """Deterministic stand-in for ``unet`` (time-conditioned 2-conv net, ~2k params)."""

    def __init__(self, c, hidden=8, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.c1 = nn.Conv2d(c, hidden, 3, padding=1)
        self.c2 = nn.Conv2d(hidden, c, 3, padding=1)
        self.te = nn.Linear(1, hidden)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)

    def forward(self, x, t, return_dict=False, **kw):
        h = self.c1(x) + self.te(t.float()[:, None] / 1000.0)[:, :, None, None]
        return (self.c2(torch.tanh(h)),)
