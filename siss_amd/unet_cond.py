"""Stable-Diffusion UNet (diffusers ``UNet2DConditionModel``, v1.x) forward + dual-cotangent backward on HIP.

SURVEY.md §8 a-U "SD UNet (config 5)" / Appendix A7: the network ``delete_sd.py:977-985`` reaches through
``unet(x_mix, t, encoder_hidden_states=..., return_dict=False)[0]`` (losses/ddpm_deletion_loss.py:24).  The
resnet / down- and up-sampling / GroupNorm / head machinery is inherited from :class:`UNetEngine`; this file adds
the ``Transformer2DModel`` block in TOKEN space (compact bf16 rows ``[B*H*W][C]``):

    GroupNorm(eps 1e-6) -> proj_in (1x1) -> { LN -> self-attn -> +res ; LN -> cross-attn(text) -> +res ;
                                              LN -> GEGLU feed-forward -> +res } -> proj_out (1x1) -> + input

Every matrix product (projections, QK^T, PV, feed-forward, and all their dgrad / wgrad counterparts) runs on the
MFMA GEMMs of gemm_nt.hip / gemm_tn.hip; LayerNorm, GEGLU, head split/merge and the row softmax are the HBM-bound
kernels of transformer.hip.  Attention is evaluated per (sample, head) as batched GEMMs over head-split, zero-padded
operands ``[B*heads][S_pad][D_pad]`` with the probability matrix kept for the backward (288 GB of HBM: at 64x64
latents it is 32 MiB per (sample, head) in bf16).  The backward carries both SISS cotangent sets at once
(rows ``[0, B*S)`` seed g_x, ``[B*S, 2*B*S)`` seed g_a) against the B saved samples, like the rest of the engine.

Parameter names are the diffusers state-dict keys.
"""
import torch

from . import lib, ops
from .config import UNet2DConditionConfig
from .layout import Act
from .unet import UNetEngine


def _up(n, m):
    return -(-n // m) * m


class UNetCondEngine(UNetEngine):
    # The side-stream weight gradients (UNetEngine.wgrad_side) pay at small batches only: this network's low-resolution stretch is
    # transformer blocks whose fused attention kernels (D = 160: one 4-wave block per CU, whole register file) run 3-5x longer on
    # the half of the chip the side launch leaves.  Same box, alternating: B = 4 46.24 / 46.27 ms with, 46.54 / 46.66 without;
    # B = 16 110.06 / 110.04 with, 108.88 / 108.90 without.
    side_max_batch = 8
    # Round 6: ONE stream for this network.  With the sparse gradient fill the side work beside the forward pass is 0.85 ms of traffic and the
    # side-stream weight gradients were neutral (above), while a captured step that forks to a second stream takes the runtime's per-node
    # graph launch: 32.5 ms of host time per 42.6-ms step at B = 4 (78 of 105 at B = 16) against 0.27 ms for the linear graph -- same step
    # time on the device (same box: B = 4 42.60 / 42.56 linear, 42.56 / 42.88 forked; B = 16 105.18 / 105.17, 104.99 / 105.53;
    # tools/probes/graph_linear.sh), and a host that is no longer within a quarter of becoming the bound.
    wgrad_side = False
    prep_side = False
    fuse_geglu_bwd = True      # GEGLU backward in the epilogue of the producing dgrad product (round 6)
    fuse_geglu_fwd = True      # GEGLU forward in the epilogue of its input projection (round 6)
    fuse_kv = True             # cross-attention: to_k / to_v as one [2C][Ckv] projection and one weight-gradient product (round 6)

    def __init__(self, cfg: UNet2DConditionConfig, device="cuda", dtype=torch.bfloat16):
        super().__init__(cfg, device, dtype=dtype)
        self.ctx = None
        # fused attention (csrc/flash_attn.hip) for the transformer blocks; False: batched GEMMs + row softmax with the
        # S x S matrices in HBM (the round-1 form: the parity tests compare the two; the f32 parity mode runs it -- the fused
        # kernels are bf16-only)
        self.flash = not self.f32

    # ------------------------------------------------------------------ parameters
    def _declare_transformer(self, pre, ch):
        a, X = self.ps.add, self.cfg.cross_attention_dim
        a(f"{pre}.norm.weight", "vec", (ch,)); a(f"{pre}.norm.bias", "vec", (ch,))
        a(f"{pre}.proj_in.weight", "conv1", (ch, ch, 1, 1)); a(f"{pre}.proj_in.bias", "vec", (ch,))
        b = f"{pre}.transformer_blocks.0"
        for i, kv in ((1, ch), (2, X)):
            a(f"{b}.norm{i}.weight", "vec", (ch,)); a(f"{b}.norm{i}.bias", "vec", (ch,))
            a(f"{b}.attn{i}.to_q.weight", "mat", (ch, ch))
            a(f"{b}.attn{i}.to_k.weight", "mat", (ch, kv))
            a(f"{b}.attn{i}.to_v.weight", "mat", (ch, kv))
            a(f"{b}.attn{i}.to_out.0.weight", "mat", (ch, ch)); a(f"{b}.attn{i}.to_out.0.bias", "vec", (ch,))
        a(f"{b}.norm3.weight", "vec", (ch,)); a(f"{b}.norm3.bias", "vec", (ch,))
        a(f"{b}.ff.net.0.proj.weight", "mat", (8 * ch, ch)); a(f"{b}.ff.net.0.proj.bias", "vec", (8 * ch,))
        a(f"{b}.ff.net.2.weight", "mat", (ch, 4 * ch)); a(f"{b}.ff.net.2.bias", "vec", (ch,))
        a(f"{pre}.proj_out.weight", "conv1", (ch, ch, 1, 1)); a(f"{pre}.proj_out.bias", "vec", (ch,))

    def _declare_params(self):
        cfg, a = self.cfg, self.ps.add
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        self.temb_dim = temb
        a("conv_in.weight", "conv_in", (ch[0], cfg.in_channels, 3, 3)); a("conv_in.bias", "vec", (ch[0],))
        a("time_embedding.linear_1.weight", "mat", (temb, ch[0])); a("time_embedding.linear_1.bias", "vec", (temb,))
        a("time_embedding.linear_2.weight", "mat", (temb, temb)); a("time_embedding.linear_2.bias", "vec", (temb,))
        self.plan_down, self.plan_up = [], []
        out = ch[0]
        for i, kind in enumerate(cfg.down_block_types):
            cin, out = out, ch[i]
            attn, down = kind.startswith("CrossAttn"), i != len(ch) - 1
            for j in range(cfg.layers_per_block):
                self._declare_resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb)
                if attn:
                    self._declare_transformer(f"down_blocks.{i}.attentions.{j}", out)
            if down:
                a(f"down_blocks.{i}.downsamplers.0.conv.weight", "conv3", (out, out, 3, 3))
                a(f"down_blocks.{i}.downsamplers.0.conv.bias", "vec", (out,))
            self.plan_down.append((i, cin, out, attn, down))
        c = ch[-1]
        self._declare_resnet("mid_block.resnets.0", c, c, temb)
        self._declare_transformer("mid_block.attentions.0", c)
        self._declare_resnet("mid_block.resnets.1", c, c, temb)
        rev = list(reversed(ch))
        out = rev[0]
        n = cfg.layers_per_block + 1
        for i, kind in enumerate(cfg.up_block_types):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            attn, up = kind.startswith("CrossAttn"), i != len(ch) - 1
            rs = []
            for j in range(n):
                skip = cin if j == n - 1 else out
                rin = prev if j == 0 else out
                self._declare_resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb)
                rs.append((rin, skip))
                if attn:
                    self._declare_transformer(f"up_blocks.{i}.attentions.{j}", out)
            if up:
                a(f"up_blocks.{i}.upsamplers.0.conv.weight", "conv3", (out, out, 3, 3))
                a(f"up_blocks.{i}.upsamplers.0.conv.bias", "vec", (out,))
            self.plan_up.append((i, out, attn, up, rs))
        a("conv_norm_out.weight", "vec", (ch[0],)); a("conv_norm_out.bias", "vec", (ch[0],))
        a("conv_out.weight", "conv3", (cfg.out_channels, ch[0], 3, 3)); a("conv_out.bias", "vec", (cfg.out_channels,))

    def init_random(self, seed=0, std=0.02):
        import math
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for n, sp in self.ps.specs.items():
            if sp.kind == "vec":
                if n.endswith((".norm.weight", "norm1.weight", "norm2.weight", "norm3.weight", "conv_norm_out.weight")):
                    sd[n] = torch.ones(sp.ref_shape) + 0.05 * torch.randn(sp.ref_shape, generator=g)
                else:
                    sd[n] = 0.02 * torch.randn(sp.ref_shape, generator=g)
            else:
                fan_in = math.prod(sp.ref_shape[1:])
                sd[n] = torch.randn(sp.ref_shape, generator=g) / math.sqrt(fan_in)
        self.load_state_dict(sd)
        return sd

    # ------------------------------------------------------------------ token-space primitives
    def _wgrad_sig(self):
        return super()._wgrad_sig() + (self.fuse_kv,)

    def _linear(self, x, wname, out, rows, n_out, k_in, bias=True, residual=None):
        """out[rows, n_out] = x[rows, k_in] W^T (+ b) (+ residual)."""
        ps = self.ps
        ops.gemm_nt(lib.ptr(x), k_in, ps.sh(wname + ".weight"), lib.ptr(out), n_out, rows, n_out, k_in, [0], [0],
                    bias=ps.p(wname + ".bias") if bias else None,
                    res_ptr=lib.ptr(residual) if residual is not None else None, ldr=n_out)

    def _linear_bwd(self, dy, xin, wname, rows2, rows_x, n_out, k_in, dx_out=None, accumulate=False, bias=True):
        """dy [rows2, n_out]: cotangent of y = xin W^T + b, xin [rows_x, k_in] shared by the sets when
        rows_x < rows2.  dW (+ db) for every set; dx_out (+)= dy W."""
        ps, gb, ns = self.ps, self.gbase, self.nsets
        rps = rows2 // ns
        dW = ps.grads[gb:, ps.specs[wname + ".weight"].off:]
        tiles = (-(-n_out // 128)) * (-(-k_in // 128))
        zp = ops.zero_page(self.device)
        if self.group_attn and self.group_rows:
            # queued for a grouped launch (UNetEngine._flush_wgrads): dy lives in a per-site buffer (see transformer())
            z9 = (lib.I * 9)(*([0] * 9))
            self._wq.append((lib.TNJob(Y=dy.data_ptr(), ldy=n_out, X=xin.data_ptr(), ldx=k_in, dW=dW.data_ptr(),
                                       set_stride=ps.total, N=n_out, C=k_in, npanels=1, nsets=ns, rows_per_set=rps,
                                       row_begin=0, row_end=rps, nsplits=self._ns_auto(), x_set_rows=rps if rows_x == rows2 else 0,
                                       zero_page=zp.data_ptr(),
                                       dbias=ps.g(wname + ".bias", gb).data_ptr() if bias else None, dbias2=None,
                                       shifts=z9, coffs=z9), (dy, xin)))
            if len(self._wq) >= self.group_max:
                self._flush_wgrads()
        else:
            lib.call("siss_gemm_tn", dy, n_out, xin, k_in, dW, ps.total, n_out, k_in, 1, lib.int_array([0]),
                     lib.int_array([0]), ns, rps, rps if rows_x == rows2 else 0, 0, rps,
                     ops._nsplits(tiles, 1, ns, rps, False), zp, ps.g(wname + ".bias", gb) if bias else None, None)
        if dx_out is not None:
            ops.gemm_nt(lib.ptr(dy), n_out, self.wT[wname + ".weight"], lib.ptr(dx_out), k_in, rows2, k_in, n_out,
                        [0], [0], res_ptr=lib.ptr(dx_out) if accumulate else None, ldr=k_in)

    def _layernorm(self, x, pre, nm, rows, C):
        ps = self.ps
        y = self._buf(nm + ".y", (rows, C), self.adt)
        mean, rstd = self._buf(nm + ".mean", (rows,)), self._buf(nm + ".rstd", (rows,))
        lib.call("siss_layernorm_fwd", x, ps.p(pre + ".weight"), ps.p(pre + ".bias"), y, mean, rstd, rows, C, 1e-5)

        def bwd(dy, accum, dx, rows2):
            lib.call("siss_layernorm_bwd", dy, x, ps.p(pre + ".weight"), mean, rstd, accum, dx,
                     ps.g(pre + ".weight", self.gbase), ps.g(pre + ".bias", self.gbase), rows2, rows,
                     rows2 // self.nsets, ps.total, C)
        return y, bwd

    def _attention(self, xq, xkv, pre, nm, B, Sq, Sk, C, Ckv, residual):
        """Multi-head attention  out = residual + to_out(softmax(q k^T / sqrt(d)) v), q from xq [B*Sq, C],
        k / v from xkv [B*Sk, Ckv].  Returns (out, bwd)."""
        Hh = self.cfg.heads
        D = C // Hh
        assert D * Hh == C and D % 8 == 0, f"head_dim {D} must be a multiple of 8"
        Dp, Sqp, Skp = _up(D, 64), _up(Sq, 64), _up(Sk, 64)
        BH = B * Hh
        scale = D ** -0.5
        bb = lambda s, shape, dt=None: self._buf(nm + s, shape, dt or self.adt)      # saved for the backward
        # scratch shared by all sites -- or, when the weight-gradient products are queued for grouped launches, per site
        # (a queued product reads its cotangent operand long after the next site would have reused the buffer)
        tb = lambda s, shape, dt=None: self._buf((nm if self.group_attn and self.group_rows else "tfm") + ".scr" + s, shape, dt or self.adt)
        sb = lambda s, shape, dt=None: self._buf("tfm" + s, shape, dt or self.adt)   # always shared: the S x S matrices of the materialised path
        rq, rk = B * Sq, B * Sk
        flash = self.flash and Dp in (64, 128, 192)
        # fused path: the kernels address q / k / v / o in the projections' own [rows, C] layout (head h at columns h * D): the
        # projection outputs are what the backward keeps -- no head-split / head-merge copies, no padded tensors
        keep = bb if flash else tb
        ps = self.ps
        wq, wk, wv = (ps.sh(pre + n + ".weight") for n in (".to_q", ".to_k", ".to_v"))
        # self-attention on the fused path: q, k and v are ONE projection (the three weights lie back to back in the flat
        # buffer: a [3C][C] matrix) into one [rows, 3C] tensor -- the normalised input is read once, and the attention kernels
        # take the three column blocks by pointer + row stride
        # fused path: the query leaves its projection already multiplied by softmax_scale * log2(e) (the product's alpha, applied in
        # f32 before the one rounding to bf16): the attention kernels' scores are base-2 logits as they come, one vector multiply
        # less per score element in loops that are bound by vector issue (q_prescaled = 1; dq is still d / d(unscaled q))
        qmul = float(scale) * 1.4426950408889634
        fused_qkv = (flash and xq is xkv and Ckv == C and wk.data_ptr() == wq.data_ptr() + 2 * C * C
                     and wv.data_ptr() == wk.data_ptr() + 2 * C * C)
        # cross-attention on the fused path: k and v are ONE projection of the text embedding (to_k / to_v back to back: a [2C][Ckv]
        # matrix) into one [rows, 2C] tensor, and one weight-gradient product in the backward pass (the text takes no gradient)
        fused_kv = (flash and self.fuse_kv and xq is not xkv and wv.data_ptr() == wk.data_ptr() + 2 * C * Ckv)
        if fused_qkv:
            qkv = bb(".qkv", (rq, 3 * C))
            ops.gemm_nt(lib.ptr(xq), C, wq, lib.ptr(qkv), 3 * C, rq, 3 * C, C, [0], [0], alpha=qmul, alpha_cols=C)
            q, k, v, ldqkv = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], 3 * C
            ldq = ldqkv
        else:
            q, ldq = keep(".q", (rq, C)), C
            if flash:
                ops.gemm_nt(lib.ptr(xq), C, wq, lib.ptr(q), C, rq, C, C, [0], [0], alpha=qmul)
            else:
                self._linear(xq, pre + ".to_q", q, rq, C, C, bias=False)
            if fused_kv:
                kv = keep(".kv", (rk, 2 * C))
                self._linear(xkv, pre + ".to_k", kv, rk, 2 * C, Ckv, bias=False)
                k, v, ldqkv = kv[:, :C], kv[:, C:], 2 * C
            else:
                k, v, ldqkv = keep(".k", (rk, C)), keep(".v", (rk, C)), C
                self._linear(xkv, pre + ".to_k", k, rk, C, Ckv, bias=False)
                self._linear(xkv, pre + ".to_v", v, rk, C, Ckv, bias=False)
        o = bb(".o", (rq, C))
        if flash:
            # QK^T -> softmax -> .V in ONE kernel (csrc/flash_attn.hip): the S x S matrices never reach HBM; the base-2
            # log-sum-exp is all the backward needs besides q, k, v, o
            lse = bb(".lse", (BH, Sqp), torch.float32)
            lib.call("siss_flash_attn_fwd_merged", q, ldq, k, ldqkv, v, ldqkv, o, C, lse, B, Hh, Sq, Sk, D, float(scale), 1)
        else:
            qh, kh, vh = bb(".qh", (BH, Sqp, Dp)), bb(".kh", (BH, Skp, Dp)), bb(".vh", (BH, Skp, Dp))
            lib.call("siss_head_split", q, qh, B, Sq, Hh, D, Sqp, Dp)
            lib.call("siss_head_split", k, kh, B, Sk, Hh, D, Skp, Dp)
            lib.call("siss_head_split", v, vh, B, Sk, Hh, D, Skp, Dp)
            oh = bb(".oh", (BH, Sqp, Dp))                # kept: delta = rowsum(dO o O) in the backward
            vT = sb(".vT", (BH, Dp, Skp))
            lib.call("siss_transpose_bf16", vh, vT, BH, Skp, Dp)
            sc, p = sb(".sc", (BH, Sqp, Skp)), bb(".p", (BH, Sqp, Skp))
            ops.gemm_nt(lib.ptr(qh), Dp, kh, lib.ptr(sc), Skp, Sqp, Skp, Dp, [0], [0], alpha=scale, batch=BH,
                        stride_a=Sqp * Dp, stride_w=Skp * Dp, stride_c=Sqp * Skp)
            lib.call("siss_softmax_rows_fwd", sc, p, BH * Sqp, Sk, Skp, 0)
            ops.gemm_nt(lib.ptr(p), Skp, vT, lib.ptr(oh), Dp, Sqp, Dp, Skp, [0], [0], batch=BH,
                        stride_a=Sqp * Skp, stride_w=Dp * Skp, stride_c=Sqp * Dp)
            lib.call("siss_head_merge", oh, o, B, Sq, Hh, D, Sqp, Dp)
        out = bb(".out", (rq, C))
        self._linear(o, pre + ".to_out.0", out, rq, C, C, residual=residual)

        def bwd(dout, rows2, dxq, dxkv):
            """dout [rows2, C] cotangent of `out` (the residual branch is the caller's).  dxq = dq W_q (+ for
            self-attention: dk W_k + dv W_v) is written to dxq; dxkv None: the keys / values come from the text
            embedding, which takes no gradient."""
            nb = rows2 // Sq
            nBH = nb * Hh
            zp = ops.zero_page(self.device)
            do = tb(".do", (rows2, C))
            self._linear_bwd(dout, o, pre + ".to_out.0", rows2, rq, C, C, dx_out=do)
            if fused_qkv:
                dqkv = tb(".dqkv", (rows2, 3 * C))
                dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]
            elif fused_kv:
                assert dxkv is None
                dq, dkv = tb(".dq", (rows2, C)), tb(".dkv", (nb * Sk, 2 * C))
                dk, dv = dkv[:, :C], dkv[:, C:]
            else:
                dq, dk, dv = tb(".dq", (rows2, C)), tb(".dk", (nb * Sk, C)), tb(".dv", (nb * Sk, C))
            delta = tb(".delta", (nBH * Sqp,), torch.float32)
            if flash:
                # FlashAttention-2 style: P is recomputed per tile from q, k and the saved log-sum-exp; all cotangent
                # (batch, head) entries in one launch pair, cotangent batch b against forward batch b % B; delta[q] =
                # sum_k P[q][k] dP[q][k] = <dO[q], O[q]> is formed by the dQ kernel from tiles it loads anyway
                lib.call("siss_flash_attn_bwd_merged", q, ldq, k, ldqkv, v, ldqkv, o, C, do, C, lse, delta, dq, ldq, dk, ldqkv,
                         dv, ldqkv, nb, B, Hh, Sq, Sk, D, float(scale), 1)
            else:
                doh = tb(".doh", (nBH, Sqp, Dp))
                lib.call("siss_head_split", do, doh, nb, Sq, Hh, D, Sqp, Dp)
                dqh = tb(".dqh", (nBH, Sqp, Dp))
                dkh, dvh = tb(".dkh", (nBH, Skp, Dp)), tb(".dvh", (nBH, Skp, Dp))
                # delta[q] = sum_k P[q][k] dP[q][k] = <dO[q], O[q]> : no pass over the S x S matrices needed for it
                lib.call("siss_rowdot", doh, oh, delta, nBH * Sqp, BH * Sqp, Dp)
                ds = sb(".ds", (nBH, Sqp, Skp))
                dkf, dvf = sb(".dkf", (nBH, Skp, Dp), torch.float32), sb(".dvf", (nBH, Skp, Dp), torch.float32)
                khT = sb(".khT", (BH, Dp, Skp))
                lib.call("siss_transpose_bf16", kh, khT, BH, Skp, Dp)
                i0, i1 = lib.int_array([0]), lib.int_array([0])
                for g in range(nb // B):             # cotangent groups that share the B forward samples
                    sl = slice(g * BH, (g + 1) * BH)
                    # dS = scale * P o (dO V^T - delta) straight from the product's epilogue: dP is never materialised
                    lib.call("siss_gemm_nt_mulsub", doh[sl], Dp, vh, ds[sl], Skp, p, Skp, delta[g * BH * Sqp:], Sqp, Skp, Dp,
                             float(scale), BH, Sqp * Dp, Skp * Dp, Sqp * Skp)
                    # dV[key][d] = sum_q P[q][key] dO[q][d]
                    lib.call("siss_gemm_tn", p, Skp, doh[sl], Dp, dvf[sl], Skp * Dp, Skp, Dp, 1, i0, i1, BH, Sqp, Sqp,
                             0, Sqp, -1, zp, None, None)     # -1: one split, dW overwritten (no zero fill)
                for g in range(nb // B):
                    sl = slice(g * BH, (g + 1) * BH)
                    # dQ = dS K
                    ops.gemm_nt(lib.ptr(ds[sl]), Skp, khT, lib.ptr(dqh[sl]), Dp, Sqp, Dp, Skp, [0], [0], batch=BH,
                                stride_a=Sqp * Skp, stride_w=Dp * Skp, stride_c=Sqp * Dp)
                    # dK[key][d] = sum_q dS[q][key] Q[q][d]
                    lib.call("siss_gemm_tn", ds[sl], Skp, qh, Dp, dkf[sl], Skp * Dp, Skp, Dp, 1, i0, i1, BH, Sqp, Sqp,
                             0, Sqp, -1, zp, None, None)     # -1: one split, dW overwritten (no zero fill)
                lib.call("siss_cast_f32_bf16", dkf, dkh, dkf.numel())
                lib.call("siss_cast_f32_bf16", dvf, dvh, dvf.numel())
                lib.call("siss_head_merge", dqh, dq, nb, Sq, Hh, D, Sqp, Dp)
                lib.call("siss_head_merge", dkh, dk, nb, Sk, Hh, D, Skp, Dp)
                lib.call("siss_head_merge", dvh, dv, nb, Sk, Hh, D, Skp, Dp)
            if fused_qkv:
                # one weight-gradient product for the [3C][C] matrix (cotangent rows [rows2, 3C]) and one three-panel product
                # dx = dq W_q + dk W_k + dv W_v (panel p: cotangent columns [pC, (p+1)C) against the p-th transposed copy)
                self._linear_bwd(dqkv, xq, pre + ".to_q", rows2, rq, 3 * C, C, dx_out=None, bias=False)
                wts = [self.wT[pre + n + ".weight"] for n in (".to_q", ".to_k", ".to_v")]
                assert wts[1].data_ptr() == wts[0].data_ptr() + 2 * C * C and wts[2].data_ptr() == wts[1].data_ptr() + 2 * C * C
                ops.gemm_nt(lib.ptr(dqkv), 3 * C, wts[0], lib.ptr(dxq), C, rows2, C, C, [0, 0, 0], [0, C, 2 * C])
            elif fused_kv:
                self._linear_bwd(dq, xq, pre + ".to_q", rows2, rq, C, C, dx_out=dxq, bias=False)
                self._linear_bwd(dkv, xkv, pre + ".to_k", nb * Sk, rk, 2 * C, Ckv, dx_out=None, bias=False)
            else:
                self._linear_bwd(dq, xq, pre + ".to_q", rows2, rq, C, C, dx_out=dxq, bias=False)
                self._linear_bwd(dk, xkv, pre + ".to_k", nb * Sk, rk, C, Ckv, dx_out=dxkv, accumulate=True, bias=False)
                self._linear_bwd(dv, xkv, pre + ".to_v", nb * Sk, rk, C, Ckv, dx_out=dxkv, accumulate=True, bias=False)
        return out, bwd

    def transformer(self, x: Act, pre):
        """Transformer2DModel with one BasicTransformerBlock (Appendix A7)."""
        ps = self.ps
        C, B, S = x.c, x.n, x.h * x.w
        X, Sk = self.cfg.cross_attention_dim, self.ctx_len
        assert self.ctx is not None and self.ctx.shape[0] == B * Sk, "encoder_hidden_states batch mismatch"
        rows = B * S
        nm = self._name(pre)
        b = pre + ".transformer_blocks.0"
        bb = lambda s, shape, dt=None: self._buf(nm + s, shape, dt or self.adt)
        hn, gn_b = self.gn(x, pre + ".norm", False, compact_out=True, eps=1e-6)
        x0 = bb(".x0", (rows, C))
        self._linear(hn, pre + ".proj_in", x0, rows, C, C)
        n1, ln1_b = self._layernorm(x0, b + ".norm1", nm + ".ln1", rows, C)
        x1, at1_b = self._attention(n1, n1, b + ".attn1", nm + ".at1", B, S, S, C, C, residual=x0)
        n2, ln2_b = self._layernorm(x1, b + ".norm2", nm + ".ln2", rows, C)
        x2, at2_b = self._attention(n2, self.ctx, b + ".attn2", nm + ".at2", B, S, Sk, C, X, residual=x1)
        n3, ln3_b = self._layernorm(x2, b + ".norm3", nm + ".ln3", rows, C)
        hff = bb(".hff", (rows, 8 * C))
        gg = bb(".gg", (rows, 4 * C))
        if self.fuse_geglu_fwd and (4 * C) % 64 == 0 and not self.f32 and lib.has("siss_gemm_nt_geglu_fwd"):
            # the projection h = [a | g] and a * gelu(g) from ONE launch (the tile holds matching value / gate columns)
            lib.call("siss_gemm_nt_geglu_fwd", n3, C, ps.sh(b + ".ff.net.0.proj.weight"), ps.p(b + ".ff.net.0.proj.bias"), hff, gg,
                     rows, 4 * C, C)
        else:
            self._linear(n3, b + ".ff.net.0.proj", hff, rows, 8 * C, C)
            lib.call("siss_geglu_fwd", hff, gg, rows, 4 * C)
        x3 = bb(".x3", (rows, C))
        self._linear(gg, b + ".ff.net.2", x3, rows, C, 4 * C, residual=x2)
        y = bb(".y", (rows, C))
        self._linear(x3, pre + ".proj_out", y, rows, C, C)
        out = self._act(nm + ".out", B, x.h, x.w, C)
        lib.call("siss_compact_add_to_pad", y, x.data, out.data, B, x.h, x.w, C)

        def bwd():
            nb = self.nb
            rows2 = nb * S
            dout = self._take(out)
            tb = lambda s, shape, dt=None: self._buf((nm if self.group_attn and self.group_rows else "tfm") + ".scr" + s, shape, dt or self.adt)
            dy = tb(".dy", (rows2, C))
            lib.call("siss_pad_to_compact", dout.data, dy, nb, x.h, x.w, C)
            dx3 = tb(".dx3", (rows2, C))
            self._linear_bwd(dy, x3, pre + ".proj_out", rows2, rows, C, C, dx_out=dx3)
            # feed-forward
            dhff = tb(".dhff", (rows2, 8 * C))
            if self.fuse_geglu_bwd and not self.f32 and lib.has("siss_gemm_nt_geglu_bwd"):
                # the output projection's dgrad with the GEGLU backward in its epilogue (siss_gemm_nt_geglu_bwd): the [rows2, 4 C]
                # cotangent of the GEGLU output never reaches HBM (4 of the 12 bytes per element the two launches moved)
                self._linear_bwd(dx3, gg, b + ".ff.net.2", rows2, rows, C, 4 * C, dx_out=None)
                lib.call("siss_gemm_nt_geglu_bwd", dx3, C, self.wT[b + ".ff.net.2.weight"], dhff, hff, rows, rows2, 4 * C, C)
            else:
                dgg = tb(".dgg", (rows2, 4 * C))
                self._linear_bwd(dx3, gg, b + ".ff.net.2", rows2, rows, C, 4 * C, dx_out=dgg)
                lib.call("siss_geglu_bwd", dgg, hff, dhff, rows2, rows, 4 * C)
            dn = tb(".dn", (rows2, C))
            self._linear_bwd(dhff, n3, b + ".ff.net.0.proj", rows2, rows, 8 * C, C, dx_out=dn)
            dx2 = tb(".dx2", (rows2, C))
            ln3_b(dn, dx3, dx2, rows2)                         # dx2 = LN3^T dn + dx3 (residual)
            # cross-attention (keys / values from the text embedding: no cotangent to propagate)
            at2_b(dx2, rows2, dn, None)
            dx1 = tb(".dx1", (rows2, C))
            ln2_b(dn, dx2, dx1, rows2)
            # self-attention
            at1_b(dx1, rows2, dn, dn)
            dx0 = tb(".dx0", (rows2, C))
            ln1_b(dn, dx1, dx0, rows2)
            dhn = tb(".dhn", (rows2, C))
            self._linear_bwd(dx0, hn, pre + ".proj_in", rows2, rows, C, C, dx_out=dhn)
            dx = gn_b(dhn, accum=dout)                         # the block's residual: d_out passes straight through
            self._give(x, dx)
        self.tape.append(bwd)
        return out

    # UNetEngine.forward calls self.attention(h, "<block>.attentions.<j>") at the attention sites
    attention = transformer

    # ------------------------------------------------------------------ whole network
    def forward(self, x, t, encoder_hidden_states=None):
        """x: [N, 4, H, W] latents, t: [N] int64, encoder_hidden_states: [N, L, cross_attention_dim]
        (delete_sd.py:941-976).  Returns pred [N, 4, H, W] f32."""
        assert encoder_hidden_states is not None, "UNet2DConditionModel needs encoder_hidden_states"
        e = encoder_hidden_states
        assert e.dim() == 3 and e.shape[0] == x.shape[0] and e.shape[2] == self.cfg.cross_attention_dim
        self.ctx_len = e.shape[1]
        ctx = self._buf("ctx", (e.shape[0] * e.shape[1], e.shape[2]), self.adt)
        ctx.copy_(e.reshape(-1, e.shape[2]))
        self.ctx = ctx
        return super().forward(x, t)
