"""CLIP text encoder forward on HIP -- the frozen ``text_encoder`` of the SD checkpoint that turns the prompt's token
ids into the UNet's ``encoder_hidden_states`` (delete_sd.py:469-474 load, :941-944
``text_encoder(input_ids, return_dict=False)[0]``; SURVEY.md §8f rank 4, the step before the path).

Forward only (the module is frozen: ``requires_grad_(False)``, delete_sd.py:476-478).  Token space, compact bf16 rows
``[B*S][C]``: LayerNorm / causal row softmax / quick_gelu are the HBM-bound kernels of transformer.hip, every product
(q/k/v/out projections, QK^T, PV, fc1, fc2) runs on the MFMA NT GEMM with bias and residual fused in its epilogue.
Weights are kept as bf16 operand copies (f32 biases and LayerNorm parameters), keyed by the transformers state-dict
names (``text_model.`` prefix optional, as it differs between transformers 4.x and 5.x).
"""
import torch

from . import lib, ops


def _up(n, m):
    return -(-n // m) * m


class CLIPTextEncoder:
    def __init__(self, state_dict, num_attention_heads=12, layer_norm_eps=1e-5, device="cuda"):
        lib.load()
        self.device = torch.device(device)
        lib.ensure_workspace(self.device)
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()
              if "position_ids" not in k}
        self.tok = sd["embeddings.token_embedding.weight"].to(self.device, torch.float32)
        self.pos = sd["embeddings.position_embedding.weight"].to(self.device, torch.float32)
        self.C = self.tok.shape[1]
        self.heads, self.eps = int(num_attention_heads), float(layer_norm_eps)
        assert self.C % self.heads == 0 and (self.C // self.heads) % 8 == 0 and self.C % 64 == 0
        self.n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
        self.w, self.f = {}, {}
        for k, v in sd.items():
            if k.startswith("embeddings."):
                continue
            if k.endswith("_proj.weight") or k.endswith("fc1.weight") or k.endswith("fc2.weight"):
                self.w[k] = v.to(self.device, torch.bfloat16).contiguous()          # GEMM operand copy [out][in]
            else:
                self.f[k] = v.to(self.device, torch.float32).contiguous()           # biases, LayerNorm gamma / beta
        self.inner = self.w["encoder.layers.0.mlp.fc1.weight"].shape[0]
        assert self.inner % 64 == 0
        self._bufs = {}

    @classmethod
    def from_pretrained(cls, path, subfolder="text_encoder", device="cuda"):
        """diffusers / transformers on-disk layout: <path>/<subfolder>/{config.json, model.safetensors}."""
        import json
        import os
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        cfg = json.load(open(os.path.join(d, "config.json")))
        return cls(load_file(os.path.join(d, "model.safetensors")), cfg.get("num_attention_heads", 12),
                   cfg.get("layer_norm_eps", 1e-5), device)

    def _buf(self, name, shape, dtype=torch.bfloat16):
        k = (name, tuple(shape), dtype)
        b = self._bufs.get(k)
        if b is None:
            n = 1
            for s in shape:
                n *= s
            b = torch.zeros(n + 1024, dtype=dtype, device=self.device)[:n].view(shape)
            self._bufs[k] = b
        return b

    def _linear(self, x, name, out, rows, n_out, k_in, residual=None):
        ops.gemm_nt(lib.ptr(x), k_in, self.w[name + ".weight"], lib.ptr(out), n_out, rows, n_out, k_in, [0], [0],
                    bias=self.f[name + ".bias"], res_ptr=lib.ptr(residual) if residual is not None else None, ldr=n_out)

    def _ln(self, x, name, out, rows):
        mean, rstd = self._buf("mean", (rows,), torch.float32), self._buf("rstd", (rows,), torch.float32)
        lib.call("siss_layernorm_fwd", x, self.f[name + ".weight"], self.f[name + ".bias"], out, mean, rstd, rows,
                 self.C, self.eps)

    @torch.no_grad()
    def __call__(self, input_ids, return_dict=False):
        """input_ids [B, S] int64 (S <= max_position_embeddings).  Returns (last_hidden_state [B, S, C] f32,)."""
        ids = input_ids.to(self.device)
        B, S = ids.shape
        C, Hh = self.C, self.heads
        D = C // Hh
        Dp, Sp = _up(D, 64), _up(S, 64)
        rows, BH = B * S, B * Hh
        bb = lambda n, shape, dt=torch.bfloat16: self._buf(n, shape, dt)
        x, y = bb("x", (rows, C)), bb("y", (rows, C))
        x.copy_((self.tok[ids] + self.pos[:S]).reshape(rows, C))          # embedding gather: plumbing
        h = bb("h", (rows, C))
        q, k, v = bb("q", (rows, C)), bb("k", (rows, C)), bb("v", (rows, C))
        qh, kh, vh = bb("qh", (BH, Sp, Dp)), bb("kh", (BH, Sp, Dp)), bb("vh", (BH, Sp, Dp))
        vT, sc, p = bb("vT", (BH, Dp, Sp)), bb("sc", (BH, Sp, Sp)), bb("p", (BH, Sp, Sp))
        oh, o = bb("oh", (BH, Sp, Dp)), bb("o", (rows, C))
        f1 = bb("f1", (rows, self.inner))
        for i in range(self.n_layers):
            pre = f"encoder.layers.{i}"
            self._ln(x, pre + ".layer_norm1", h, rows)
            for nm, dst, hd in (("q_proj", q, qh), ("k_proj", k, kh), ("v_proj", v, vh)):
                self._linear(h, f"{pre}.self_attn.{nm}", dst, rows, C, C)
                lib.call("siss_head_split", dst, hd, B, S, Hh, D, Sp, Dp)
            lib.call("siss_transpose_bf16", vh, vT, BH, Sp, Dp)
            ops.gemm_nt(lib.ptr(qh), Dp, kh, lib.ptr(sc), Sp, Sp, Sp, Dp, [0], [0], alpha=D ** -0.5, batch=BH,
                        stride_a=Sp * Dp, stride_w=Sp * Dp, stride_c=Sp * Sp)
            lib.call("siss_softmax_rows_fwd", sc, p, BH * Sp, S, Sp, Sp)         # causal: key <= query
            ops.gemm_nt(lib.ptr(p), Sp, vT, lib.ptr(oh), Dp, Sp, Dp, Sp, [0], [0], batch=BH,
                        stride_a=Sp * Sp, stride_w=Dp * Sp, stride_c=Sp * Dp)
            lib.call("siss_head_merge", oh, o, B, S, Hh, D, Sp, Dp)
            self._linear(o, pre + ".self_attn.out_proj", y, rows, C, C, residual=x)       # y = x + attn
            self._ln(y, pre + ".layer_norm2", h, rows)
            self._linear(h, pre + ".mlp.fc1", f1, rows, self.inner, C)
            lib.call("siss_quick_gelu", f1, f1, f1.numel())
            self._linear(f1, pre + ".mlp.fc2", x, rows, C, self.inner, residual=y)        # x = y + mlp
        self._ln(x, "final_layer_norm", h, rows)
        out = h.float().view(B, S, C).clone()
        if return_dict:
            return type("BaseModelOutput", (), {"last_hidden_state": out})()
        return (out,)
