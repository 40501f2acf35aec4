"""Torch-only restatement of ``transformers.CLIPTextModel`` (oracle, test infrastructure only).

The reference encodes ONE prompt with the frozen CLIP text encoder of the SD checkpoint and feeds the result to the
UNet as ``encoder_hidden_states`` (delete_sd.py:469-474 load, :941-944 ``text_encoder(input_ids, return_dict=False)[0]``).
``transformers`` is a pip dependency of the reference (environment.yml:288, 4.38.2), not vendored under
/root/reference; it IS installed in this image, so this restatement is **pinned**: tests/test_oracle_frontend.py
checks it against ``transformers.CLIPTextModel`` on the same weights and token ids.  Parameter names are the
transformers state-dict keys (``text_model.`` prefix as in 4.38).
"""
from dataclasses import dataclass

import torch
import torch.nn as nn


@dataclass
class CLIPTextCfg:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5

    @staticmethod
    def sd_v1():
        """openai/clip-vit-large-patch14 text tower = text_encoder/config.json of SD v1.x."""
        return CLIPTextCfg()

    @staticmethod
    def tiny():
        return CLIPTextCfg(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                           num_attention_heads=2)


class _Attn(nn.Module):
    def __init__(self, c, heads):
        super().__init__()
        self.heads = heads
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x, mask):
        b, s, c = x.shape
        d = c // self.heads

        def split(t):
            return t.view(b, s, self.heads, d).transpose(1, 2)
        q, k, v = split(self.q_proj(x) * d ** -0.5), split(self.k_proj(x)), split(self.v_proj(x))
        p = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)
        return self.out_proj((p @ v).transpose(1, 2).reshape(b, s, c))


class _MLP(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(c, inner), nn.Linear(inner, c)

    def forward(self, x):
        h = self.fc1(x)
        return self.fc2(h * torch.sigmoid(1.702 * h))          # quick_gelu


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = _Attn(cfg.hidden_size, cfg.num_attention_heads)
        self.layer_norm1 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(cfg.hidden_size, cfg.intermediate_size)
        self.layer_norm2 = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)

    def forward(self, x, mask):
        x = x + self.self_attn(self.layer_norm1(x), mask)
        return x + self.mlp(self.layer_norm2(x))


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


class _TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class OracleCLIPText(nn.Module):
    def __init__(self, cfg: CLIPTextCfg):
        super().__init__()
        self.cfg = cfg
        self.text_model = _TextModel(cfg)

    def forward(self, input_ids, return_dict=False):
        """input_ids [B, S] int64 -> (last_hidden_state [B, S, C],)  -- causal self-attention, final LayerNorm."""
        tm = self.text_model
        b, s = input_ids.shape
        x = tm.embeddings.token_embedding(input_ids) + tm.embeddings.position_embedding.weight[:s]
        mask = torch.full((s, s), float("-inf"), dtype=x.dtype).triu(1)
        for layer in tm.encoder.layers:
            x = layer(x, mask)
        return (tm.final_layer_norm(x),)
