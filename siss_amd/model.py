"""Drop-in ``UNet2DModel`` object over the HIP engine (the ``unet`` contract of SURVEY.md §8b):
callable ``unet(sample, timesteps, return_dict=False)[0]`` with an autograd-connected output,
``named_parameters() / parameters() / train() / eval() / device / config``,
``from_pretrained`` / ``save_pretrained`` in the diffusers on-disk format
(``config.json`` + ``diffusion_pytorch_model.safetensors``; delete_celeb.py:137-147, :181-186).

This is the compatibility surface: the reference's own loop (two ``backward`` calls with
``retain_graph=True``, per-parameter ``.grad`` bookkeeping, torch.optim) runs against it
unchanged.  The fast path is ``siss_amd.step.SISSStepper`` (one dual-cotangent backward).
"""
import json
import os

import torch

from .config import UNet2DConditionConfig, UNet2DConfig
from .unet import UNetEngine


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, sample, timesteps, *cond):
        ctx.model = model
        ctx.token = model._fwd_token = object()
        ctx.inputs = (sample.contiguous(), timesteps, *cond)
        return model.engine.forward(*ctx.inputs).clone()

    @staticmethod
    def backward(ctx, gout):
        m = ctx.model
        eng = m.engine
        if ctx.token is not m._fwd_token:
            # another forward ran since (e.g. double_forward_with_neg_del): the engine keeps ONE set of saved
            # activations, so recompute this forward before differentiating it (checkpoint-style).
            eng.forward(*ctx.inputs)
            m._fwd_token = ctx.token
        eng.ps.grads[0].zero_()
        eng.backward(gout.contiguous().float(), nsets=1)
        m._accumulate_param_grads()
        return (None,) * len(ctx.needs_input_grad)


class UNet2DModel:
    config_cls = UNet2DConfig
    class_name = "UNet2DModel"

    @staticmethod
    def _make_engine(config, device, dtype=torch.bfloat16):
        return UNetEngine(config, device, dtype=dtype)

    def _cond_args(self, kwargs):
        """Positional conditioning tensors of the engine's forward, from the reference's **conditioning."""
        return ()

    def __init__(self, config=None, device="cuda", compute_dtype=torch.bfloat16, **kwargs):
        """compute_dtype: torch.bfloat16 (bf16 MFMA operands: `mixed_precision: bf16`, the fast path) or torch.float32 (the f32
        parity mode: `mixed_precision: null`, the reference's shipped default -- every tensor and product f32)."""
        if config is None:
            config = self.config_cls.from_dict(kwargs) if kwargs else self.config_cls()
        self.config = config
        self.engine = self._make_engine(config, device, compute_dtype)
        self.device = self.engine.device
        self.dtype = torch.float32
        self.training = True
        self._anchor = torch.zeros(1, device=self.device, requires_grad=True)
        self._fwd_token = None
        ps = self.engine.ps
        self._params = {n: torch.nn.Parameter(ps.p(n)) for n in ps.specs}   # views of the flat master (native layout)

    # ---- nn.Module-like surface -------------------------------------------------
    def named_parameters(self):
        return iter(self._params.items())

    def parameters(self):
        return iter(self._params.values())

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def requires_grad_(self, flag=True):
        for p in self._params.values():
            p.requires_grad_(flag)
        return self

    def _accumulate_param_grads(self):
        ps = self.engine.ps
        for n, p in self._params.items():
            sp = ps.specs[n]
            g = ps.grads[0, sp.off:sp.off + sp.numel].view(sp.native_shape)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad += g

    def __call__(self, sample, timestep, *args, return_dict=False, **kwargs):
        if args:                                  # diffusers' positional encoder_hidden_states
            kwargs.setdefault("encoder_hidden_states", args[0])
        cond = self._cond_args(kwargs)
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long)
        t = t.to(self.device).reshape(-1).expand(sample.shape[0]) if t.numel() == 1 else t.to(self.device)
        # master weights may have been updated in place by an external optimizer: refresh operand copies
        self.engine.refresh_weights(cast_shadow=True)
        sample = sample.to(self.device)
        if sample.dtype not in (torch.float32, torch.bfloat16):
            sample = sample.float()
        if torch.is_grad_enabled():
            out = _UNetFn.apply(self._anchor, self, sample, t, *cond)
        else:
            out = self.engine.forward(sample.contiguous(), t, *cond).clone()
        if return_dict:
            return type("UNet2DOutput", (), {"sample": out})()
        return (out,)

    # ---- checkpoints ------------------------------------------------------------
    def state_dict(self):
        return self.engine.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.engine.load_state_dict(sd, strict)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, device="cuda", compute_dtype=torch.bfloat16, **unused):
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        if not os.path.exists(os.path.join(d, "config.json")) and os.path.exists(os.path.join(path, "unet")):
            d = os.path.join(path, "unet")
        m = cls(cls.config_cls.from_json(os.path.join(d, "config.json")), device=device, compute_dtype=compute_dtype)
        m.load_state_dict(load_file(os.path.join(d, "diffusion_pytorch_model.safetensors")))
        return m

    def save_pretrained(self, d):
        from safetensors.torch import save_file
        os.makedirs(d, exist_ok=True)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.config).items()}
        cfg["_class_name"] = self.class_name
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()},
                  os.path.join(d, "diffusion_pytorch_model.safetensors"))


class UNet2DConditionModel(UNet2DModel):
    """Stable-Diffusion UNet surface: ``unet(x, t, encoder_hidden_states=..., return_dict=False)[0]``
    (delete_sd.py:458-462 load, :977-985 -> losses/ddpm_deletion_loss.py:24 call)."""
    config_cls = UNet2DConditionConfig
    class_name = "UNet2DConditionModel"

    @staticmethod
    def _make_engine(config, device, dtype=torch.bfloat16):
        from .unet_cond import UNetCondEngine
        return UNetCondEngine(config, device, dtype=dtype)

    def _cond_args(self, kwargs):
        e = kwargs.get("encoder_hidden_states")
        if e is None:
            raise TypeError("UNet2DConditionModel needs encoder_hidden_states")
        return (e.to(self.device),)
