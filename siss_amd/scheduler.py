"""DDPM noise schedule (the part of diffusers' DDPMScheduler the unlearning loop uses:
delete_celeb.py:229 load, :367-371 alphas_cumprod -> gamma/sigma, :602-603 add_noise;
config/train_tshirt_mnist.yaml:43-50 for the initialise-from-config form)."""
import json
import os

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, beta_schedule="linear",
                 prediction_type="epsilon", **unused):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise ValueError(f"unsupported beta_schedule {beta_schedule!r}")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.config = type("Cfg", (), dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                           beta_end=beta_end, beta_schedule=beta_schedule,
                                           prediction_type=prediction_type))()

    @classmethod
    def from_pretrained(cls, path, subfolder="scheduler"):
        fn = os.path.join(path, subfolder, "scheduler_config.json")
        with open(fn) as f:
            d = json.load(f)
        return cls(**{k: v for k, v in d.items() if not k.startswith("_")})

    def add_noise(self, original_samples, noise, timesteps):
        """alphas_cumprod is cast to the sample dtype first (bf16 mode rounds it) -- SURVEY Appendix A1.
        The fused HIP kernel (csrc/siss_loss.hip) reproduces this bit-for-bit; this torch form exists
        for the class-surface path where callers ask for noisy latents explicitly."""
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        a = (ac[timesteps] ** 0.5).flatten()
        b = ((1 - ac[timesteps]) ** 0.5).flatten()
        while a.dim() < original_samples.dim():
            a, b = a.unsqueeze(-1), b.unsqueeze(-1)
        return a * original_samples + b * noise


def lr_multiplier(name, step, num_warmup_steps=0, num_training_steps=0):
    """LR multiplier at scheduler step `step` of diffusers.optimization.get_scheduler(name, ...) (0.27.2, as published):
    what delete_celeb.py:296-301 builds from cfg.lr_scheduler / cfg.warmup_steps / cfg.training_steps and steps after
    every optimizer update (:770).  Schedules outside this table raise instead of silently training at a constant rate."""
    import math
    w, total = int(num_warmup_steps), int(num_training_steps)
    if name == "constant":
        return 1.0
    if name == "constant_with_warmup":
        return step / max(1.0, w) if step < w else 1.0
    if name == "linear":
        if step < w:
            return step / max(1, w)
        return max(0.0, (total - step) / max(1, total - w))
    if name == "cosine":
        if step < w:
            return step / max(1, w)
        progress = (step - w) / max(1, total - w)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    raise NotImplementedError(f"lr_scheduler={name!r}: implemented: constant, constant_with_warmup, linear, cosine "
                              "(diffusers.optimization.get_scheduler)")
