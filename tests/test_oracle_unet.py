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


def test_mnist_config_one_siss_step_on_cpu():
    """BASELINE config 1 (delete_tshirt.yaml: MNIST 28x28 DDPM, SISS, CPU reference path): one optimizer step of
    the oracle at a reduced batch -- plumbing check of the CPU path at the real architecture."""
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    torch.manual_seed(0)
    net = OracleUNet2D(UNetConfig.mnist_tshirt())
    ac = S.alphas_cumprod()
    opt = torch.optim.AdamW(net.parameters(), lr=5e-5, betas=(0.95, 0.999), weight_decay=1e-6)
    g = torch.Generator().manual_seed(46)
    B = 2
    mb = dict(x0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1, a0=torch.rand(B, 1, 28, 28, generator=g) * 2 - 1,
              noise=torch.randn(B, 1, 28, 28, generator=g), t=torch.randint(0, 1000, (B,), generator=g),
              u=torch.rand(B, generator=g))
    before = [p.detach().clone() for p in net.parameters()]
    st, gx, ga, gfin = unlearning_step(net, opt, OracleDeletionLoss(*S.gamma_sigma(ac)),
                                       "importance_sampling_with_mixture", ac, [mb], train_batch_size=B,
                                       scaling_norm=5.0, loss_params={"lambd": 0.5}, inf_guard=True)
    assert all(map(lambda v: v == v and v != float("inf"), (st.norm_loss_x, st.norm_loss_a, st.pre_clip_norm)))
    assert abs(st.scaling_factor * st.norm_loss_a - 5.0) < 1e-4          # norm fixing: ||s * g_a|| = scaling_norm
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))


def test_sd15_param_count_and_keys():
    """UNet2DConditionModel of runwayml/stable-diffusion-v1-5 (config/delete_sd.yaml:70): 859,520,964 parameters in
    686 tensors, diffusers key names."""
    from oracle.unet_cond import OracleUNet2DCondition, UNetCondConfig
    with torch.device("meta"):
        m = OracleUNet2DCondition(UNetCondConfig.sd15())
    ps = dict(m.named_parameters())
    assert sum(p.numel() for p in ps.values()) == 859_520_964
    assert len(ps) == 686
    for k in ("conv_in.weight", "time_embedding.linear_2.bias", "down_blocks.0.attentions.0.norm.weight",
              "down_blocks.0.attentions.1.proj_in.weight",
              "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight",
              "down_blocks.2.attentions.1.transformer_blocks.0.attn2.to_k.weight",
              "mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.bias",
              "up_blocks.1.attentions.2.transformer_blocks.0.ff.net.2.weight",
              "up_blocks.3.attentions.2.proj_out.bias", "up_blocks.2.upsamplers.0.conv.weight",
              "down_blocks.3.resnets.1.conv2.weight", "conv_out.bias"):
        assert k in ps, k
    assert "down_blocks.3.attentions.0.norm.weight" not in ps and "up_blocks.0.attentions.0.norm.weight" not in ps
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.bias" not in ps
    assert ps["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_v.weight"].shape == (320, 768)
    assert ps["mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight"].shape == (10240, 1280)
    assert ps["up_blocks.1.resnets.0.conv1.weight"].shape == (1280, 2560, 3, 3)
    assert ps["up_blocks.3.resnets.2.conv1.weight"].shape == (320, 640, 3, 3)


def test_sd_tiny_forward_backward():
    from oracle.unet_cond import OracleUNet2DCondition, UNetCondConfig
    torch.manual_seed(0)
    m = OracleUNet2DCondition(UNetCondConfig.tiny())
    x = torch.randn(2, 4, 16, 16)
    y = m(x, torch.tensor([3, 999]), torch.randn(2, 7, 64))[0]
    assert y.shape == x.shape
    y.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
