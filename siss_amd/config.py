"""UNet2DModel architecture description (the keys of a diffusers ``unet/config.json``).

The reference loads the network with ``DDPMPipeline.from_pretrained`` / ``UNet2DModel``
(delete_celeb.py:181-186, delete_tshirt.py:180-183; config/*.yaml ``unet._target_:
diffusers.UNet2DModel``); this dataclass carries the same fields so a checkpoint's
config.json can be read unchanged.
"""
import json
from dataclasses import dataclass, fields
from typing import Optional, Tuple


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
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names}
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
        return UNet2DConditionConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names})

    @staticmethod
    def from_json(path):
        with open(path) as f:
            return UNet2DConditionConfig.from_dict(json.load(f))

    @property
    def heads(self):
        return self.attention_head_dim
