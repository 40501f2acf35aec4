"""Pins for the UNet restatement (parity unpinned by the reference: diffusers is
un-vendored).  Exact parameter / tensor counts of the public checkpoints and the
diffusers state-dict key names (SURVEY.md §8 a-U)."""
import torch

from oracle.unet import OracleUNet2D, UNetConfig


def test_celebahq_param_count_and_keys():
    m = OracleUNet2D(UNetConfig.celebahq256())
    ps = dict(m.named_parameters())
    assert sum(p.numel() for p in ps.values()) == 113_673_219
    assert len(ps) == 450
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "time_embedding.linear_2.bias",
              "down_blocks.0.resnets.0.norm1.weight", "down_blocks.0.resnets.1.time_emb_proj.weight",
              "down_blocks.0.downsamplers.0.conv.weight", "down_blocks.2.resnets.0.conv_shortcut.weight",
              "down_blocks.4.attentions.1.to_out.0.bias", "down_blocks.4.attentions.0.group_norm.weight",
              "mid_block.attentions.0.to_q.weight", "mid_block.resnets.1.conv2.weight",
              "up_blocks.0.resnets.2.conv_shortcut.weight", "up_blocks.1.attentions.2.to_v.weight",
              "up_blocks.0.upsamplers.0.conv.weight", "conv_norm_out.weight", "conv_out.bias"):
        assert k in ps, k
    assert "down_blocks.5.downsamplers.0.conv.weight" not in ps
    assert "up_blocks.5.upsamplers.0.conv.weight" not in ps
    assert ps["up_blocks.0.resnets.0.conv1.weight"].shape == (512, 1024, 3, 3)
    assert ps["up_blocks.5.resnets.2.conv1.weight"].shape == (128, 256, 3, 3)


def test_mnist_param_count_and_forward():
    m = OracleUNet2D(UNetConfig.mnist_tshirt())
    assert sum(p.numel() for p in m.parameters()) == 14_735_745
    x = torch.randn(2, 1, 28, 28)
    y = m(x, torch.tensor([0, 999]), return_dict=False)[0]
    assert y.shape == x.shape and torch.isfinite(y).all()


def test_tiny_forward_backward():
    torch.manual_seed(0)
    m = OracleUNet2D(UNetConfig.tiny())
    x = torch.randn(2, 3, 16, 16)
    y = m(x, torch.tensor([3, 999]))[0]
    y.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
