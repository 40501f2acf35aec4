"""DDPM noise schedule restatement (oracle; test infrastructure only).

Restates diffusers==0.27.2 ``DDPMScheduler`` (pinned in the reference's
environment.yml:232, not vendored).  Call sites in the reference:
delete_celeb.py:229 (load), :367-371 (gamma/sigma), :602-603 (add_noise).
"""
import torch


def make_betas(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02,
               beta_schedule="linear"):
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                              dtype=torch.float32) ** 2
    raise ValueError(beta_schedule)


def alphas_cumprod(**kw):
    return torch.cumprod(1.0 - make_betas(**kw), dim=0)


def gamma_sigma(ac):
    """delete_celeb.py:367-371: gamma = sqrt(abar), sigma = sqrt(1 - abar)."""
    return ac ** 0.5, (1 - ac) ** 0.5


def add_noise(ac, x, noise, t):
    """DDPMScheduler.add_noise: alphas_cumprod is cast to the SAMPLE dtype first
    (so in bf16 mode gamma/sigma used for noising come from bf16-rounded abar)."""
    ac = ac.to(device=x.device, dtype=x.dtype)
    a = ac[t] ** 0.5
    b = (1 - ac[t]) ** 0.5
    a = a.flatten()
    b = b.flatten()
    while a.dim() < x.dim():
        a = a.unsqueeze(-1)
        b = b.unsqueeze(-1)
    return a * x + b * noise
