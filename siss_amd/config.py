"""UNet2DModel architecture description (the keys of a diffusers ``unet/config.json``).

The reference loads the network with ``DDPMPipeline.from_pretrained`` / ``UNet2DModel``
(delete_celeb.py:181-186, delete_tshirt.py:180-183; config/*.yaml ``unet._target_:
diffusers.UNet2DModel``); this dataclass carries the same fields so a checkpoint's
config.json can be read unchanged.
"""
import json
from dataclasses import dataclass, fields
from typing import Optional, Tuple

# Keys of a diffusers 0.27.2 `unet/config.json` that are NOT fields of the dataclasses below, with the only values the
# HIP networks implement.  A checkpoint that carries anything else (scale-shift time conditioning, Fourier time
# embedding, no attention, a centred input, ...) describes a DIFFERENT network: loading it must fail, not compute
# something else silently.
_FIXED_2D = {
    "center_input_sample": (False,), "time_embedding_type": ("positional",), "mid_block_scale_factor": (1, 1.0),
    "downsample_type": ("conv",), "upsample_type": ("conv",), "dropout": (0, 0.0), "attn_norm_num_groups": (None,),
    "resnet_time_scale_shift": ("default",), "add_attention": (True,), "class_embed_type": (None,),
    "num_class_embeds": (None,), "num_train_timesteps": (None,),
}
_FIXED_COND = {
    "center_input_sample": (False,), "time_embedding_type": ("positional",), "mid_block_scale_factor": (1, 1.0),
    "mid_block_type": ("UNetMidBlock2DCrossAttn",), "only_cross_attention": (False,), "dropout": (0, 0.0),
    "transformer_layers_per_block": (1,), "reverse_transformer_layers_per_block": (None,), "encoder_hid_dim": (None,),
    "encoder_hid_dim_type": (None,), "num_attention_heads": (None,), "dual_cross_attention": (False,),
    "use_linear_projection": (False,), "class_embed_type": (None,), "addition_embed_type": (None,),
    "addition_time_embed_dim": (None,), "num_class_embeds": (None,), "upcast_attention": (False,),
    "resnet_time_scale_shift": ("default",), "resnet_skip_time_act": (False,), "resnet_out_scale_factor": (1, 1.0),
    "time_embedding_dim": (None,), "time_embedding_act_fn": (None,), "timestep_post_act": (None,),
    "time_cond_proj_dim": (None,), "conv_in_kernel": (3,), "conv_out_kernel": (3,),
    "projection_class_embeddings_input_dim": (None,), "attention_type": ("default",), "class_embeddings_concat": (False,),
    "mid_block_only_cross_attention": (None,), "cross_attention_norm": (None,), "addition_embed_type_num_heads": (64,),
}


def _checked(d, names, fixed, what, down_ok, up_ok):
    """Split a config dict into dataclass fields; raise on keys / values the implementation does not cover."""
    kw = {}
    for k, v in d.items():
        if k.startswith("_"):                       # _class_name, _diffusers_version, _name_or_path, hydra's _target_
            continue
        if k in names:
            kw[k] = tuple(v) if isinstance(v, list) else v
        elif k in fixed:
            if v not in fixed[k]:
                raise ValueError(f"{what} config: {k}={v!r} is not implemented (supported: {fixed[k][0]!r}); "
                                 "refusing to load a checkpoint of a different architecture")
        else:
            raise ValueError(f"{what} config: unknown key {k!r}")
    if kw.get("act_fn", "silu") not in ("silu", "swish"):
        raise ValueError(f"{what} config: act_fn={kw['act_fn']!r} is not implemented (silu)")
    for key, ok in (("down_block_types", down_ok), ("up_block_types", up_ok)):
        bad = [b for b in kw.get(key, ()) if b not in ok]
        if bad:
            raise ValueError(f"{what} config: {key} {bad} not implemented (supported: {sorted(ok)})")
    return kw


@dataclass
class UNet2DConfig:
    sample_size: int = 256
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 128, 256, 256, 512, 512)
    down_block_types: Tuple[str, ...] = ("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4
    layers_per_block: int = 2
    attention_head_dim: Optional[int] = None
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    downsample_padding: int = 0
    flip_sin_to_cos: bool = False
    freq_shift: int = 1
    act_fn: str = "silu"

    @staticmethod
    def celebahq256():
        """google/ddpm-celebahq-256 (config/delete_celeb.yaml:7)."""
        return UNet2DConfig()

    @staticmethod
    def mnist_tshirt():
        """config/train_tshirt_mnist.yaml:25-41 + UNet2DModel defaults."""
        return UNet2DConfig(sample_size=28, in_channels=1, out_channels=1, block_out_channels=(64, 128, 256),
                            down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
                            up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                            attention_head_dim=8, norm_eps=1e-5, downsample_padding=1,
                            flip_sin_to_cos=True, freq_shift=0)

    @staticmethod
    def from_dict(d):
        names = {f.name for f in fields(UNet2DConfig)}
        kw = _checked(d, names, _FIXED_2D, "UNet2DModel", {"DownBlock2D", "AttnDownBlock2D"}, {"UpBlock2D", "AttnUpBlock2D"})
        # UNet2DModel defaults that differ from the celeb dataclass defaults
        kw.setdefault("attention_head_dim", 8)
        kw.setdefault("norm_eps", 1e-5)
        kw.setdefault("downsample_padding", 1)
        kw.setdefault("flip_sin_to_cos", True)
        kw.setdefault("freq_shift", 0)
        return UNet2DConfig(**kw)

    @staticmethod
    def from_json(path):
        with open(path) as f:
            return UNet2DConfig.from_dict(json.load(f))

    def head_dim(self, channels):
        return self.attention_head_dim if self.attention_head_dim is not None else channels


@dataclass
class UNet2DConditionConfig:
    """``UNet2DConditionModel`` (Stable Diffusion v1.x UNet; delete_sd.py:458-462, config/delete_sd.yaml:70).
    diffusers quirk kept as is: for this model ``attention_head_dim`` is the NUMBER of heads (8)."""
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    attention_head_dim: int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    downsample_padding: int = 1
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    act_fn: str = "silu"

    @staticmethod
    def sd15():
        return UNet2DConditionConfig()

    @staticmethod
    def from_dict(d):
        names = {f.name for f in fields(UNet2DConditionConfig)}
        return UNet2DConditionConfig(**_checked(d, names, _FIXED_COND, "UNet2DConditionModel",
                                                {"DownBlock2D", "CrossAttnDownBlock2D"}, {"UpBlock2D", "CrossAttnUpBlock2D"}))

    @staticmethod
    def from_json(path):
        with open(path) as f:
            return UNet2DConditionConfig.from_dict(json.load(f))

    @property
    def heads(self):
        return self.attention_head_dim
