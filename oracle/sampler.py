"""DDPM ancestral sampling restated on CPU (oracle; test infrastructure only).

Restates what the reference's evaluation path uses of diffusers==0.27.2 (un-vendored, so **parity
unpinned by the reference**): ``DDPMScheduler.set_timesteps`` (leading spacing), ``DDPMScheduler.step``
(epsilon prediction, fixed_small variance, clip_sample) and the ``DDPMPipeline`` loop.  Reference call
sites: evaluate.py:37-50 (sample_images, 50 steps) and evaluate.py:64-79 (inject at t, then t+1 single steps).
"""
import torch


def inference_timesteps(num_train, num_inference):
    ratio = num_train // num_inference
    return [int(round(i * ratio)) for i in range(num_inference)][::-1]


def step_coeffs(ac, t, num_train, num_inference=None):
    """Scalars of one reverse step x_t -> x_{t-1} (all float64 python numbers)."""
    n = num_inference if num_inference else num_train
    prev_t = t - num_train // n
    a_t = float(ac[t])
    a_prev = float(ac[prev_t]) if prev_t >= 0 else 1.0
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    c_x0 = (a_prev ** 0.5) * cur_b / b_t
    c_xt = (cur_a ** 0.5) * b_prev / b_t
    var = max(b_prev / b_t * cur_b, 1e-20) if t > 0 else 0.0
    return dict(sqrt_a=a_t ** 0.5, sqrt_b=b_t ** 0.5, c_x0=c_x0, c_xt=c_xt, sigma=var ** 0.5)


def ddpm_step(ac, eps, t, x, noise, num_train=1000, num_inference=None, clip=True):
    c = step_coeffs(ac, t, num_train, num_inference)
    x0 = (x - c["sqrt_b"] * eps) / c["sqrt_a"]
    if clip:
        x0 = x0.clamp(-1, 1)
    out = c["c_x0"] * x0 + c["c_xt"] * x
    if t > 0:
        out = out + c["sigma"] * noise
    return out


def sample(unet, ac, x_T, noises, num_inference, num_train=1000):
    """DDPMPipeline loop with caller-supplied noise (noises[i] is used at the i-th step)."""
    x = x_T
    with torch.no_grad():
        for i, t in enumerate(inference_timesteps(num_train, num_inference)):
            eps = unet(x, torch.full((x.shape[0],), t, dtype=torch.long), return_dict=False)[0]
            x = ddpm_step(ac, eps, t, x, noises[i], num_train, num_inference)
    return (x / 2 + 0.5).clamp(0, 1)


def denoise(unet, ac, x_t, noises, timestep, num_train=1000):
    """evaluate.py:64-79: t = timestep, timestep-1, ..., 0 with single-step spacing."""
    x = x_t
    with torch.no_grad():
        for i, t in enumerate(reversed(range(timestep + 1))):
            eps = unet(x, torch.full((x.shape[0],), t, dtype=torch.long), return_dict=False)[0]
            x = ddpm_step(ac, eps, t, x, noises[i], num_train, None)
    return ((x + 1) / 2).clamp(0, 1)
