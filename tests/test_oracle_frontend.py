"""Pins for the restatements of the SD front end (oracle/clip_text.py): checked against the installed
``transformers.CLIPTextModel`` -- the reference's own text encoder class (delete_sd.py:469-474) -- on identical weights."""
import pytest
import torch


def _hf_state(model):
    """transformers >= 5 drops the ``text_model.`` prefix of 4.38's state dict; normalise to the 4.38 names."""
    sd = {}
    for k, v in model.state_dict().items():
        if "position_ids" in k:
            continue
        sd[k if k.startswith("text_model.") else "text_model." + k] = v
    return sd


@pytest.mark.parametrize("case", ["tiny", "sd_width_2_layers"])
def test_clip_text_restatement_matches_transformers(case):
    tf = pytest.importorskip("transformers")
    from oracle.clip_text import CLIPTextCfg, OracleCLIPText
    cfg = CLIPTextCfg.tiny() if case == "tiny" else CLIPTextCfg(vocab_size=2000, num_hidden_layers=2)
    hf_cfg = tf.CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                               intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                               num_attention_heads=cfg.num_attention_heads,
                               max_position_embeddings=cfg.max_position_embeddings, hidden_act="quick_gelu",
                               layer_norm_eps=cfg.layer_norm_eps, bos_token_id=0, eos_token_id=cfg.vocab_size - 1)
    torch.manual_seed(0)
    hf = tf.CLIPTextModel(hf_cfg).eval()
    mine = OracleCLIPText(cfg).eval()
    sd = _hf_state(hf)
    assert set(sd) == set(mine.state_dict())
    mine.load_state_dict(sd)
    ids = torch.randint(0, cfg.vocab_size, (3, 77), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf(ids, return_dict=False)[0]
        got = mine(ids)[0]
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)


def test_clip_text_sd_v1_parameter_count():
    """The SD v1.x text encoder (CLIP ViT-L/14 text tower): 123,060,480 parameters."""
    from oracle.clip_text import CLIPTextCfg, OracleCLIPText
    with torch.device("meta"):
        m = OracleCLIPText(CLIPTextCfg.sd_v1())
    assert sum(p.numel() for p in m.parameters()) == 123_060_480


def test_vae_encoder_parameter_count_and_keys():
    """CompVis/stable-diffusion-v1-4 vae: encoder 34,163,592 parameters + quant_conv 72, diffusers key names."""
    from oracle.vae import OracleVAEEncoder, VAEConfig
    with torch.device("meta"):
        m = OracleVAEEncoder(VAEConfig.sd_v1())
    ps = dict(m.named_parameters())
    assert sum(p.numel() for n, p in ps.items() if n.startswith("encoder.")) == 34_163_592
    assert sum(p.numel() for n, p in ps.items() if n.startswith("quant_conv.")) == 72
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.resnets.1.conv2.bias",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "encoder.down_blocks.2.downsamplers.0.conv.weight",
              "encoder.mid_block.attentions.0.to_q.weight", "encoder.mid_block.attentions.0.to_out.0.bias",
              "encoder.mid_block.resnets.1.norm2.weight", "encoder.conv_norm_out.bias", "encoder.conv_out.weight",
              "quant_conv.weight"):
        assert k in ps, k
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in ps
    assert ps["encoder.conv_out.weight"].shape == (8, 512, 3, 3)


def test_quant_conv_folds_into_conv_out():
    """siss_amd/vae.py loads conv_out with quant_conv folded in (W' = W_q W_out, b' = W_q b_out + b_q): the same
    algebra on the CPU equals the two-convolution form of the oracle."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    w_out, b_out = torch.randn(8, 32, 3, 3, generator=g), torch.randn(8, generator=g)
    w_q, b_q = torch.randn(8, 8, 1, 1, generator=g), torch.randn(8, generator=g)
    x = torch.randn(2, 32, 9, 9, generator=g)
    ref = F.conv2d(F.conv2d(x, w_out, b_out, padding=1), w_q, b_q)
    wq = w_q[:, :, 0, 0]
    got = F.conv2d(x, torch.einsum("om,mckl->ockl", wq, w_out), wq @ b_out + b_q, padding=1)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
