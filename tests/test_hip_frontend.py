"""Parity of the SD front end on HIP (the step before the path, delete_sd.py:879-888,:941-944):
the CLIP text encoder against oracle/clip_text.py (itself pinned by transformers.CLIPTextModel,
tests/test_oracle_frontend.py) and the VAE encoder against oracle/vae.py, same weights and inputs.

Tolerances (bf16 operands / f32 accumulate vs fp32 oracles): max-abs error <= 3e-2 of the output's max magnitude.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from siss_amd import lib
    lib.load()
    return torch.device("cuda:0")


def _close(got, ref, rel, what=""):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err / scale:.3g} > {rel})"


@pytest.mark.parametrize("case", ["tiny", "sd_width_2_layers"])
def test_clip_text_encoder_matches_oracle(dev, case):
    from siss_amd.text_encoder import CLIPTextEncoder
    from oracle.clip_text import CLIPTextCfg, OracleCLIPText
    cfg = CLIPTextCfg.tiny() if case == "tiny" else CLIPTextCfg(vocab_size=2000, num_hidden_layers=2)
    torch.manual_seed(0)
    net = OracleCLIPText(cfg).eval()
    with torch.no_grad():
        for n, p in net.named_parameters():          # spread the LayerNorm parameters and biases off their defaults
            if "layer_norm" in n or n.endswith(".bias"):
                p.add_(0.1 * torch.randn_like(p))
    ids = torch.randint(0, cfg.vocab_size, (3, 77), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = net(ids)[0]
    enc = CLIPTextEncoder(net.state_dict(), cfg.num_attention_heads, cfg.layer_norm_eps, dev)
    got = enc(ids, return_dict=False)[0]
    assert got.shape == ref.shape and got.dtype == torch.float32
    _close(got.cpu(), ref, 3e-2, "clip last_hidden_state")
    # a shorter prompt batch (S < 77) reuses the encoder
    with torch.no_grad():
        ref2 = net(ids[:1, :20])[0]
    _close(enc(ids[:1, :20])[0].cpu(), ref2, 3e-2, "clip S=20")


@pytest.mark.parametrize("case", ["tiny", "sd_widths"])
def test_vae_encoder_matches_oracle(dev, case):
    from siss_amd.vae import VAEEncoder, VAEEncoderConfig
    from oracle.vae import OracleVAEEncoder, VAEConfig
    if case == "tiny":
        kw, hw = dict(block_out_channels=(64, 128), layers_per_block=1), 32
    else:
        kw, hw = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=1), 128
    torch.manual_seed(0)
    net = OracleVAEEncoder(VAEConfig(**kw)).eval()
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "norm" in n or n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, hw, hw, generator=g) * 2 - 1
    with torch.no_grad():
        mean_r, logvar_r = net.moments(x)
    enc = VAEEncoder(VAEEncoderConfig(**kw), dev)
    enc.load_state_dict(net.state_dict())
    mean, logvar = enc.moments(x.to(dev))
    down = 2 ** (len(kw["block_out_channels"]) - 1)
    assert tuple(mean.shape) == tuple(mean_r.shape) == (2, 4, hw // down, hw // down)
    scale = max(mean_r.abs().max().item(), logvar_r.abs().max().item())
    assert (mean.cpu() - mean_r).abs().max().item() <= 3e-2 * scale
    assert (logvar.cpu() - logvar_r).abs().max().item() <= 3e-2 * scale
    eps = torch.randn(mean_r.shape, generator=g)
    with torch.no_grad():
        lat_r = net.encode(x, eps)
    _close(enc.encode(x.to(dev), eps=eps).cpu(), lat_r, 3e-2, "latents")


def test_sd_v1_front_end_full_size(dev):
    """The real shapes: 512x512 images -> [B,4,64,64] latents (34.2 M-parameter encoder, 64x64 mid attention with
    4096-long softmax rows) and 77 tokens -> [B,77,768] (123 M-parameter text tower); finite, right shapes, and the
    sampled latents are mean + std * eps of the moments."""
    from siss_amd.text_encoder import CLIPTextEncoder
    from siss_amd.vae import VAEEncoder, VAEEncoderConfig
    from oracle.clip_text import CLIPTextCfg, OracleCLIPText
    from oracle.vae import OracleVAEEncoder, VAEConfig
    torch.manual_seed(0)
    vae_ref = OracleVAEEncoder(VAEConfig.sd_v1())
    assert sum(p.numel() for p in vae_ref.parameters()) == 34_163_592 + 72
    enc = VAEEncoder(VAEEncoderConfig(), dev)
    enc.load_state_dict(vae_ref.state_dict())
    x = (torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(dev)
    mean, logvar = enc.moments(x)
    assert mean.shape == (2, 4, 64, 64) and torch.isfinite(mean).all() and torch.isfinite(logvar).all()
    eps = torch.randn(2, 4, 64, 64, device=dev)
    lat = enc.encode(x, eps=eps)
    torch.testing.assert_close(lat, (mean + torch.exp(0.5 * logvar) * eps) * 0.18215, rtol=1e-5, atol=1e-6)
    clip_ref = OracleCLIPText(CLIPTextCfg.sd_v1())
    txt = CLIPTextEncoder(clip_ref.state_dict(), 12, 1e-5, dev)
    ids = torch.randint(0, 49408, (1, 77))
    h = txt(ids)[0]
    assert h.shape == (1, 77, 768) and torch.isfinite(h).all()
    # LayerNorm output: per-token mean / variance follow gamma=1, beta=0 of the fresh final_layer_norm
    assert abs(float(h.mean())) < 5e-2 and abs(float(h.var()) - 1) < 0.1


def test_delete_sd_task_with_front_end_on_disk(dev, tmp_path):
    """delete_sd.DeleteSD end to end from a checkpoint directory in the diffusers layout (unet/, vae/,
    text_encoder/): images are VAE-encoded per micro-batch, the prompt's token ids go through the text encoder once
    (delete_sd.py:879-888, :941-944), then the SISS steps run on the latents."""
    import json
    import os
    import sys
    from safetensors.torch import save_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from siss_amd import hydra_lite as H
    from siss_amd.config import UNet2DConditionConfig
    from siss_amd.model import UNet2DConditionModel
    from oracle.clip_text import CLIPTextCfg, OracleCLIPText
    from oracle.vae import OracleVAEEncoder, VAEConfig
    ckpt = tmp_path / "ckpt"
    ucfg = UNet2DConditionConfig(sample_size=16, block_out_channels=(64, 128),
                                 down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), attention_head_dim=2,
                                 cross_attention_dim=128)
    unet = UNet2DConditionModel(ucfg, device=dev)
    unet.engine.init_random(seed=3)
    unet.save_pretrained(str(ckpt / "unet"))
    torch.manual_seed(0)
    vae = OracleVAEEncoder(VAEConfig.tiny())
    os.makedirs(ckpt / "vae")
    json.dump(dict(in_channels=3, latent_channels=4, block_out_channels=[64, 128], layers_per_block=1,
                   norm_num_groups=32, scaling_factor=0.18215), open(ckpt / "vae" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in vae.state_dict().items()}, str(ckpt / "vae" / "diffusion_pytorch_model.safetensors"))
    clip = OracleCLIPText(CLIPTextCfg.tiny())
    os.makedirs(ckpt / "text_encoder")
    json.dump(dict(num_attention_heads=2, layer_norm_eps=1e-5, hidden_size=128), open(ckpt / "text_encoder" / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in clip.state_dict().items()}, str(ckpt / "text_encoder" / "model.safetensors"))
    g = torch.Generator().manual_seed(1)
    torch.save(torch.rand(8, 3, 32, 32, generator=g) * 2 - 1, tmp_path / "all.pt")
    torch.save(torch.rand(1, 3, 32, 32, generator=g) * 2 - 1, tmp_path / "del.pt")
    ids = torch.randint(0, 1000, (1, 77), generator=g)
    torch.save(ids, tmp_path / "prompt_ids.pt")
    cfg = H.compose("delete_sd", os.path.join(root, "config"),
                    ["training_steps=2", "train_batch_size=2", "gradient_accumulation_steps=1",
                     f"output_dir={tmp_path}/out", f"pretrained_model_name_or_path={ckpt}",
                     f"images_all={tmp_path}/all.pt", f"images_deletion={tmp_path}/del.pt", "save_final=false"])
    cfg.validation_prompts = [str(tmp_path / "prompt_ids.pt")]
    task = H.instantiate(cfg.task, cfg=cfg, _recursive_=False)
    stepper = task.run()
    assert task.vae is not None and task.text_encoder is not None
    lines = [json.loads(l) for l in open(os.path.join(cfg.output_dir, "train_log_rank0.jsonl"))]
    assert len(lines) == 2 and all(abs(st["scaling_factor"] * st["norm_loss_a"] - 750.0) < 1.0 for st in lines)
    # the conditioning the steps ran with is the text encoder's output for these ids
    with torch.no_grad():
        ref = clip(ids)[0]
    got = task.conditioning(2, dev)["encoder_hidden_states"]
    assert got.shape == (2, 77, 128)
    assert (got[0].cpu() - ref[0]).abs().max() <= 3e-2 * ref.abs().max()
