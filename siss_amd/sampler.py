"""Sampling / evaluation path on the HIP forward kernels: DDPM ancestral sampling
(`Evaluator.sample_images`, evaluate.py:37-50) and inject-then-denoise (`Evaluator.denoise_images`,
evaluate.py:64-79).  Forward only: the UNet engine's forward + one fused DDPM-step kernel per step.
"""
import torch

from . import lib


def _coeffs(ac, t, num_train, num_inference):
    n = num_inference if num_inference else num_train
    prev_t = t - num_train // n
    a_t = float(ac[t])
    a_prev = float(ac[prev_t]) if prev_t >= 0 else 1.0
    b_t, b_prev = 1 - a_t, 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    var = max(b_prev / b_t * cur_b, 1e-20) if t > 0 else 0.0
    return a_t ** 0.5, b_t ** 0.5, (a_prev ** 0.5) * cur_b / b_t, (cur_a ** 0.5) * b_prev / b_t, var ** 0.5


def ddpm_step(scheduler, eps, t, x, noise=None, num_inference=None, clip=True, out=None):
    """x_{t-1} from (x_t, eps) -- DDPMScheduler.step (epsilon, fixed_small, clip_sample)."""
    T = scheduler.config.num_train_timesteps
    sa, sb, c0, ct, sig = _coeffs(scheduler.alphas_cumprod, int(t), T, num_inference)
    out = torch.empty_like(x) if out is None else out
    lib.call("siss_ddpm_step", x, eps, noise if int(t) > 0 else None, out, x.numel(), sa, sb, c0, ct, sig, int(clip))
    return out


def inference_timesteps(num_train, num_inference):
    ratio = num_train // num_inference
    return [int(round(i * ratio)) for i in range(num_inference)][::-1]


class Evaluator:
    """Same method names as the reference's evaluate.Evaluator; `unet` is a siss_amd.model.UNet2DModel."""

    def __init__(self, cfg=None, use_graph=True):
        self.cfg = cfg
        self.use_graph = use_graph
        self._graphs = {}

    def load_model(self, unet, noise_scheduler):
        self.unet, self.noise_scheduler = unet, noise_scheduler
        self._graphs = {}

    def _eps(self, x, t):
        """eps = UNet(x, t).  The forward is a fixed schedule of ~230 launches whose only per-step inputs are the
        sample and the timestep, both device tensors: it is captured ONCE per batch shape into a hipGraph and
        replayed for every denoising step (at batch 1 the eager launch path costs as much as the kernels)."""
        eng = self.unet.engine
        if not self.use_graph:
            tt = torch.full((x.shape[0],), int(t), dtype=torch.long, device=x.device)
            return eng.forward(x.contiguous(), tt)
        key = tuple(x.shape)
        ent = self._graphs.get(key)
        if ent is None:
            xs = torch.zeros(x.shape, dtype=torch.float32, device=x.device)
            ts = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                eng.forward(xs, ts)                       # settle every buffer of this shape before the capture
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    out = eng.forward(xs, ts)
            torch.cuda.current_stream().wait_stream(side)
            ent = self._graphs[key] = (graph, xs, ts, out)
        graph, xs, ts, out = ent
        xs.copy_(x)
        ts.fill_(int(t))
        graph.replay()
        return out

    @torch.no_grad()
    def sample_images(self, num_samples, num_inference_steps=None, set_generator=False, x_T=None, noises=None):
        """Returns [N, H, W, C] float numpy images in [0, 1] (DDPMPipeline output_type='numpy')."""
        u, sch = self.unet, self.noise_scheduler
        dev = u.device
        steps = num_inference_steps or (self.cfg.pipeline.num_inference_steps if self.cfg else 50)
        gen = None
        if set_generator:
            gen = torch.Generator(device=dev).manual_seed(int(self.cfg.random_seed) if self.cfg else 0)
        shape = (num_samples, u.config.in_channels, u.config.sample_size, u.config.sample_size)
        x = torch.randn(shape, device=dev, generator=gen) if x_T is None else x_T.to(dev).float().contiguous()
        u.engine.refresh_weights(cast_shadow=True)
        T = sch.config.num_train_timesteps
        for i, t in enumerate(inference_timesteps(T, steps)):
            eps = self._eps(x, t)
            noise = (torch.randn(shape, device=dev, generator=gen) if noises is None else noises[i].to(dev)) if t > 0 else None
            x = ddpm_step(sch, eps, t, x, noise, num_inference=steps)
        return (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()

    @torch.no_grad()
    def denoise_images(self, noisy_image_batch, timestep, num_inference_steps=None, set_generator=True, noises=None):
        u, sch = self.unet, self.noise_scheduler
        dev = u.device
        gen = None
        if set_generator:
            gen = torch.Generator(device=dev).manual_seed(int(self.cfg.random_seed) if self.cfg else 0)
        x = noisy_image_batch.to(dev).float().contiguous()
        u.engine.refresh_weights(cast_shadow=True)
        for i, t in enumerate(reversed(range(int(timestep) + 1))):
            eps = self._eps(x, t)
            noise = (torch.randn(x.shape, device=dev, generator=gen) if noises is None else noises[i].to(dev)) if t > 0 else None
            x = ddpm_step(sch, eps, t, x, noise)
        return ((x + 1) / 2).clamp(0, 1).permute(0, 2, 3, 1)
