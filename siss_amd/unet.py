"""DDPM UNet (diffusers ``UNet2DModel`` architecture) forward + dual-cotangent backward on HIP.

This is the a-U row of SURVEY.md §8: the network the reference reaches through
``unet(x, t, return_dict=False)[0]`` (losses/ddpm_deletion_loss.py:24) and differentiates
twice with ``retain_graph=True`` (delete_celeb.py:686-711).  Here it is a static schedule of
hand-written gfx950 kernels (csrc/*.hip) over

  * ONE flat f32 master parameter buffer (+ bf16 shadow in kernel-native layouts),
  * ONE flat f32 gradient buffer with two sets  [g_x ; g_a],
  * NHWC bf16 activations with a zero halo (layout.py),

and the backward pass carries BOTH cotangents at once (batch 2B against B saved
activations): activations are read once for g_x and g_a, dgrad runs at batch 2B and the
weight-gradient GEMMs write the two sets side by side.  No autograd, no tracing: the schedule
is replayable from a hipGraph.

Parameter names are the diffusers state-dict keys, so checkpoints round-trip.
"""
import math
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from . import lib, ops
from .config import UNet2DConfig
from .layout import Act, ActView

ALIGN = 64  # floats; keeps every bf16 shadow slice 128-B aligned


@dataclass
class PSpec:
    name: str
    kind: str          # conv3 | conv_in | conv1 | mat | vec
    ref_shape: tuple
    native_shape: tuple
    off: int
    numel: int


class ParamStore:
    """Flat master parameters / gradients / bf16 shadow, with reference<->native layout maps."""

    def __init__(self):
        self.specs = {}
        self.total = 0

    def add(self, name, kind, ref_shape):
        if kind == "conv3":
            co, ci = ref_shape[:2]
            native = (9, co, ci)
        elif kind == "conv_in":
            co, ci = ref_shape[:2]
            kp = -(-9 * ci // 64) * 64
            native = (co, kp)
        elif kind == "conv1":
            native = tuple(ref_shape[:2])
        else:
            native = tuple(ref_shape)
        numel = int(math.prod(native))
        sp = PSpec(name, kind, tuple(ref_shape), native, self.total, numel)
        self.specs[name] = sp
        self.total += -(-numel // ALIGN) * ALIGN
        return sp

    def relayout(self, early_final):
        """Re-assign offsets so that the parameters whose gradients are final EARLY in the backward pass
        (early_final(name) is True) form one contiguous tail [split, total): data-parallel runs all-reduce that
        tail while the rest of the backward is still running.  Call before allocate()."""
        order = [sp for sp in self.specs.values() if not early_final(sp.name)]
        split_at = len(order)
        order += [sp for sp in self.specs.values() if early_final(sp.name)]
        off = 0
        self.split = 0
        for i, sp in enumerate(order):
            if i == split_at:
                self.split = off
            sp.off = off
            off += -(-sp.numel // ALIGN) * ALIGN
        if split_at == len(order):
            self.split = off
        self.total = off

    def allocate(self, device, nsets=2, f32=False):
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grads = torch.zeros(nsets, self.total, dtype=torch.float32, device=device)
        # the operand copy of the parameters: their bf16 rounding -- or, in the f32 parity mode, the master itself
        self.shadow = self.flat if f32 else torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        self.nsets = nsets

    # views -----------------------------------------------------------------
    def p(self, name):
        sp = self.specs[name]
        return self.flat[sp.off:sp.off + sp.numel].view(sp.native_shape)

    def sh(self, name):
        sp = self.specs[name]
        return self.shadow[sp.off:sp.off + sp.numel].view(sp.native_shape)

    def g(self, name, base_set=0):
        """gradient slice of `name` starting at set `base_set` (kernels add set*total themselves)."""
        sp = self.specs[name]
        return self.grads[base_set, sp.off:sp.off + sp.numel]

    # layout maps -----------------------------------------------------------
    @staticmethod
    def to_native(sp, t):
        t = t.detach().to(torch.float32)
        if sp.kind == "conv3":
            return ops.conv_w_to_native(t)
        if sp.kind == "conv_in":
            co, ci = t.shape[:2]
            out = torch.zeros(sp.native_shape, dtype=torch.float32, device=t.device)
            out[:, :9 * ci] = t.permute(0, 2, 3, 1).reshape(co, 9 * ci)
            return out
        if sp.kind == "conv1":
            return t.reshape(sp.native_shape)
        return t

    @staticmethod
    def from_native(sp, t):
        if sp.kind == "conv3":
            return ops.conv_w_from_native(t.view(sp.native_shape))
        if sp.kind == "conv_in":
            co, ci = sp.ref_shape[:2]
            return t.view(sp.native_shape)[:, :9 * ci].reshape(co, 3, 3, ci).permute(0, 3, 1, 2).contiguous()
        return t.reshape(sp.ref_shape).clone()

    def load_state_dict(self, sd, strict=True):
        missing = [n for n in self.specs if n not in sd]
        extra = [n for n in sd if n not in self.specs]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]} unexpected {extra[:5]}")
        for n, sp in self.specs.items():
            if n not in sd:
                continue
            t = sd[n]
            if tuple(t.shape) != sp.ref_shape:
                raise ValueError(f"{n}: shape {tuple(t.shape)} != {sp.ref_shape}")
            self.flat[sp.off:sp.off + sp.numel] = self.to_native(sp, t.to(self.flat.device)).reshape(-1)

    def state_dict(self):
        return {n: self.from_native(sp, self.flat[sp.off:sp.off + sp.numel]).cpu() for n, sp in self.specs.items()}

    def grads_ref(self, which):
        """Gradient set `which` (0 = g_x, 1 = g_a) in reference layout, {name: cpu tensor}."""
        return {n: self.from_native(sp, self.grads[which, sp.off:sp.off + sp.numel]).cpu()
                for n, sp in self.specs.items()}

    def flat_to_ref(self, flat):
        return {n: self.from_native(sp, flat[sp.off:sp.off + sp.numel]).cpu() for n, sp in self.specs.items()}


class UNetEngine:
    """Static-schedule UNet.  ``forward(x, t)`` then ``backward(c, nsets)``."""
    # Schedule switches: plain attributes, the defaults are the measured-best settings (docs/experiments.md); the parity
    # tests flip them on an instance to compare the two forms of the same math.
    epi_stats = True       # GroupNorm statistics from the producing conv's epilogue (no statistics pass at the large sites)
    d2s_epilogue = True    # downsample dgrad: depth-to-space in the plane GEMMs' epilogue (no dz tensor, no scatter pass)
    direct_cat = True      # convs that feed a concat write into the concat buffer directly (False: copy both parts)
    fold_shortcut = True   # a resnet's 1x1 conv_shortcut rides in its conv2's 3x3 product (False: own product + residual add)
    subpixel_up = True     # Upsample2D as four 2x2-tap phase convolutions on the low-resolution input (False: upsample copy + 3x3 conv)
    # ... at the sites of at least this many low-resolution pixels.  Same-box sweep (CelebA-HQ B = 16, ms per step): off 55.84 / 56.10,
    # every site 55.68, from 32 x 32 up 55.29, from 64 x 64 up 55.84 / 55.61 -- below 32 x 32 the phase products are latency-bound
    # launches that the grouped weight gradients and the slab GroupNorm of the literal form beat
    subpixel_min_px = 1024
    subpixel_queue = 1     # ... their four phase weight gradients join the grouped-wgrad queue (the tap fold follows that launch)
    side_max_batch = 1 << 30   # ... for forward batches of at most this many samples (UNetCondEngine lowers it)
    quad_stats = True      # GroupNorm statistics of tensors no persistent-conv epilogue produced: one read of the part that lacks them, kept
    #                        for the forward pass (siss_quad_stats) -- instead of a statistics pass over the whole (concat) input at every use
    phase_launch = True    # the four space-to-depth planes of a downsample dgrad / the four phases of a sub-pixel upsample forward as ONE
    #                        launch each (siss_gemm_nt_d2s_phases: the four launches' blocks, minus their split-K at the smallest sites; the f32 mode keeps the four launches)
    s2d_from_gn = True     # ... and their cotangent arrives space-to-depth from the GroupNorm backward that forms it (no layout pass)
    # Weight gradients of all but the top-resolution layers (at most group_rows reduction rows per set: CelebA-HQ's 8x8 .. 128x128
    # levels) are not launched one by one -- each alone leaves CUs idle in its last round of blocks and pays a launch's fixed
    # ~10-40 us -- but queued and run as grouped launches (siss_gemm_tn_grouped: one job table, one launch per kernel variant).
    # They only feed the flat gradient buffer, so nothing waits for them; their cotangent operand is held back from the buffer
    # pool until the group has run.  0 = off.  Swept on one box (20 k / 70 k / 280 k / 1.1 M rows): CelebA-HQ 57.03 / 56.69 /
    # 56.45 / 56.91 ms (the 256 x 256 wgrads are better off with their own one-round grids), SD v1.5 B = 16 111.0 -> 109.6 ms.
    group_rows = 280000
    group_max = 42
    group_attn = True      # ... the attention blocks' linears too
    # Single-head attention sites (attention_head_dim = None: CelebA-HQ's six) as ONE fused kernel forward and two backward
    # (csrc/attn1h.hip) between a fused q / k / v projection and to_out as a 1x1 convolution on the padded layout -- 9 launches per
    # site and step where the materialised form (False) takes 26
    fused_attn = True
    # WEIGHT GRADIENTS ON A SIDE STREAM (round 5).  Nothing in the backward pass waits for a weight gradient (they only feed the flat
    # gradient buffer), and the low-resolution middle of the pass (up / mid / down blocks at <= side_max_px pixels: grids of 13-160
    # tiles on a 256-CU chip, GroupNorm slabs, attention) leaves most CUs idle.  When the pass enters it, the grouped launches queued
    # so far (the 32 x 32 .. 128 x 128 up blocks': ~3 ms of MFMA-bound work) go to a SIDE stream, capped at side_blocks workgroups
    # (siss_gemm_tn_grouped_capped: that many CUs, the rest stays with this stream), and the grouped launches that fill up later follow
    # them there (side_follow); everything joins at the end of the pass; the cotangent operands stay out of the buffer pool until then.
    # Same-box sweeps (CelebA-HQ B = 16, ms per step; one-stream schedule 54.05-54.15): first batch only, 72 / 96 / 120 blocks:
    # 53.74 / 53.35 / 53.39; with side_follow 64 / 96 / 112 / 128 / 144 / 160 / 192 / 224 blocks: 54.60 / 53.37 / 53.01 / 52.79-52.82 /
    # 53.04 / 53.22 / 53.47 / 53.97; entering at 32 x 32 (side_max_px 1024): 53.09; smaller batches (group_max 24): 54.0.
    # What did NOT pay (docs/experiments.md, round 5): every weight gradient streamed to the side stream for the whole pass with the
    # persistent 3x3 kernel sized to the remaining CUs (59-71 ms), and slices of the queue beside each GroupNorm-backward launch
    # (54.2-56.0 ms: those launches are bound by what their resident waves keep in flight, i.e. by the CUs they lose).
    wgrad_side = True
    side_blocks = 128
    side_max_px = 256
    side_follow = 1            # the grouped launches that fill up AFTER the first batch go to the side stream too (behind it)
    prep_side = True
    # One-panel weight gradients at the TOP resolution (a resnet's 1x1 conv_shortcut, conv_out, conv_in: HBM-bound launches at
    # 200-520 TF/s) wait for the next fused 3-tap weight gradient and ride in ITS launch (siss_gemm_tn_pair: one round of blocks
    # shared by the two products; the streaming one-tap blocks run beside MFMA-bound 3-tap blocks).  Round 6 A/B (same box, CelebA-HQ
    # B = 16, alternating): 51.59 / 51.85 ms with, 51.69 / 52.12 without -- kept (docs/experiments.md).
    pair_top = True
    # Set by the stepper for the FIRST micro-batch of a step (the gradient buffer has just been zeroed and every weight receives exactly one
    # weight-gradient product per backward pass): one-split products then overwrite their tiles instead of read-add-writing them
    # (siss_gemm_tn nsplits = -2).  Off for a backward pass driven any other way (the class surface accumulates across passes).
    wgrad_overwrite = False
    fill_grads = True          # (timing probes only: a bound on what the fill costs)
    # zero_grad(sparse_key=...) skips the stretches of the gradient buffer that the coming backward pass overwrites (recorded from the
    # first pass under the same key, checked on every later one).  SD v1.5, same box: B = 4 -1.0 ms, B = 16 -0.8 ms (docs/experiments.md).
    sparse_fill = True
    sparse_min_floats = 16384  # ... stretches of at least this many floats (the rest is simply filled)
    pair_min_rows = 280000     # ... of at least this many reduction rows per set (CelebA-HQ B = 16: the 256 x 256 level)

    def __init__(self, cfg: UNet2DConfig, device="cuda", dtype=torch.bfloat16, f32_fused=False):
        """dtype = torch.float32: the f32 PARITY MODE (`mixed_precision: null` of the reference's YAMLs; csrc/f32_path.hip) -- every
        activation and operand f32, every product on the f32 MFMA, the same engine code; the fused bf16-only forms (persistent
        3x3 kernel and what rides in it, depth-to-space epilogue, grouped / side-stream wgrads, slab GroupNorm) are off.  An
        instrument (1/16 of the bf16 MFMA rate at best, and simple kernels): it pins the network to the fp32 oracle at 1e-4."""
        lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        assert dtype in (torch.bfloat16, torch.float32)
        self.adt, self.f32 = dtype, dtype == torch.float32
        # f32_fused (round 5): the f32 mode with the SCHEDULE SWITCHES of the bf16 engine left on -- folded shortcut (forward and
        # dgrad), depth-to-space epilogue, sub-pixel upsample, queued (grouped) weight gradients -- on the f32 forms of their entry
        # points (csrc/f32_path.hip): the 1e-4 pin then covers the schedule bench.py runs, not only the unfused one.  What stays off
        # in f32: the GroupNorm statistics from the conv epilogue (statistics of ROUNDED outputs: a bf16 notion), the fused
        # attention kernels, slab GroupNorm, side streams.
        self.f32_fused = bool(f32_fused) and self.f32
        if self.f32:
            self.epi_stats = False
            if not self.f32_fused:
                self.d2s_epilogue = self.fold_shortcut = False
                self.group_rows = 0
        if self.device.type == "cuda":
            lib.ensure_workspace(self.device)
        self.ps = ParamStore()
        self._declare_params()
        self.ps.relayout(self._grads_final_early)
        self.ps.allocate(self.device, f32=self.f32)
        self._build_temb_tables()
        self.wT = {}
        self._wds = {}
        self._acts = {}
        self._bufs = {}
        self._pool = {}
        self.tape = []
        self.gmap = {}
        self._uid = 0
        self.on_early_grads_final = None
        self._wq, self._held, self._held_release = [], {}, []
        self._pair1 = []
        self._wq_post = []
        self._side, self._side_busy, self._side_held, self._side_release, self._side_mark = None, False, {}, [], None
        self._side_phase = False
        self._prep_pending, self._wT_stale = False, False
        self._fill_key, self._fill_plan, self._fill_plans, self._fill_tables = None, None, {}, []
        self._up_w = {}

    # ------------------------------------------------------------------ parameters
    def _early_blocks(self):
        """Down blocks whose backward runs before the (long) high-resolution tail: the lower half of the ladder."""
        n = len(self.cfg.block_out_channels)
        return {f"down_blocks.{i}." for i in range(n // 2 + 1, n)} if n >= 4 else set()

    def _grads_final_early(self, name):
        """True for parameters whose gradient is complete once the backward pass has left the deepest down blocks
        (up blocks, mid block, output head, deep down blocks) -- except the time-embedding-fed ones, whose
        gradients come from ONE batched kernel at the very end of the backward."""
        if ".time_emb_proj." in name or name.endswith(".conv1.bias") or name.startswith(("time_embedding.", "conv_in.")):
            return False
        if name.startswith(("up_blocks.", "mid_block.", "conv_norm_out.", "conv_out.")):
            return True
        return any(name.startswith(p) for p in self._early_blocks())

    def _declare_resnet(self, pre, cin, cout, temb):
        a = self.ps.add
        if not hasattr(self, "temb_cols"):
            self.temb_cols, self.temb_ntot = {}, 0
        self.temb_cols[pre] = (self.temb_ntot, cout)          # column range in the batched time_emb_proj GEMM
        self.temb_ntot += cout
        a(f"{pre}.norm1.weight", "vec", (cin,)); a(f"{pre}.norm1.bias", "vec", (cin,))
        a(f"{pre}.conv1.weight", "conv3", (cout, cin, 3, 3)); a(f"{pre}.conv1.bias", "vec", (cout,))
        a(f"{pre}.time_emb_proj.weight", "mat", (cout, temb)); a(f"{pre}.time_emb_proj.bias", "vec", (cout,))
        a(f"{pre}.norm2.weight", "vec", (cout,)); a(f"{pre}.norm2.bias", "vec", (cout,))
        a(f"{pre}.conv2.weight", "conv3", (cout, cout, 3, 3)); a(f"{pre}.conv2.bias", "vec", (cout,))
        if cin != cout:
            a(f"{pre}.conv_shortcut.weight", "conv1", (cout, cin, 1, 1)); a(f"{pre}.conv_shortcut.bias", "vec", (cout,))

    def _build_temb_tables(self):
        """Per output column of the concatenated time_emb_proj problem: offsets (floats into the flat
        parameter buffer) of its weight row, its bias, and the conv1 bias that shares its gradient."""
        ps = self.ps
        woff = torch.empty(self.temb_ntot, dtype=torch.int64)
        boff, boff2 = torch.empty_like(woff), torch.empty_like(woff)
        for pre, (c0, cout) in self.temb_cols.items():
            cols = torch.arange(cout)
            woff[c0:c0 + cout] = ps.specs[pre + ".time_emb_proj.weight"].off + cols * self.temb_dim
            boff[c0:c0 + cout] = ps.specs[pre + ".time_emb_proj.bias"].off + cols
            boff2[c0:c0 + cout] = ps.specs[pre + ".conv1.bias"].off + cols
        self.temb_woff, self.temb_boff, self.temb_boff2 = (v.to(self.device) for v in (woff, boff, boff2))

    def _declare_attn(self, pre, ch):
        a = self.ps.add
        a(f"{pre}.group_norm.weight", "vec", (ch,)); a(f"{pre}.group_norm.bias", "vec", (ch,))
        # the three projections' weights lie back to back in the flat buffers (one [3C][C] operand: ONE fused q / k / v product, one
        # weight-gradient job, one three-panel dgrad over consecutive transposed copies), and so do their biases
        for nm in ("to_q", "to_k", "to_v"):
            a(f"{pre}.{nm}.weight", "mat", (ch, ch))
        for nm in ("to_q", "to_k", "to_v"):
            a(f"{pre}.{nm}.bias", "vec", (ch,))
        a(f"{pre}.to_out.0.weight", "mat", (ch, ch)); a(f"{pre}.to_out.0.bias", "vec", (ch,))

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
            attn, down = kind.startswith("Attn"), i != len(ch) - 1
            for j in range(cfg.layers_per_block):
                self._declare_resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb)
                if attn:
                    self._declare_attn(f"down_blocks.{i}.attentions.{j}", out)
            if down:
                a(f"down_blocks.{i}.downsamplers.0.conv.weight", "conv3", (out, out, 3, 3))
                a(f"down_blocks.{i}.downsamplers.0.conv.bias", "vec", (out,))
            self.plan_down.append((i, cin, out, attn, down))
        c = ch[-1]
        self._declare_resnet("mid_block.resnets.0", c, c, temb)
        self._declare_attn("mid_block.attentions.0", c)
        self._declare_resnet("mid_block.resnets.1", c, c, temb)
        rev = list(reversed(ch))
        out = rev[0]
        n = cfg.layers_per_block + 1
        for i, kind in enumerate(cfg.up_block_types):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            attn, up = kind.startswith("Attn"), i != len(ch) - 1
            rs = []
            for j in range(n):
                skip = cin if j == n - 1 else out
                rin = prev if j == 0 else out
                self._declare_resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb)
                rs.append((rin, skip))
                if attn:
                    self._declare_attn(f"up_blocks.{i}.attentions.{j}", out)
            if up:
                a(f"up_blocks.{i}.upsamplers.0.conv.weight", "conv3", (out, out, 3, 3))
                a(f"up_blocks.{i}.upsamplers.0.conv.bias", "vec", (out,))
            self.plan_up.append((i, out, attn, up, rs))
        a("conv_norm_out.weight", "vec", (ch[0],)); a("conv_norm_out.bias", "vec", (ch[0],))
        a("conv_out.weight", "conv3", (cfg.out_channels, ch[0], 3, 3)); a("conv_out.bias", "vec", (cfg.out_channels,))

    def init_random(self, seed=0, std=0.02):
        """Random-init weights of the exact architecture (no checkpoints in the repo): conv/linear
        ~ N(0, 1/fan_in) scaled, GroupNorm gamma=1 beta=0, biases small."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for n, sp in self.ps.specs.items():
            if sp.kind == "vec":
                if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("group_norm.weight") \
                        or n == "conv_norm_out.weight":
                    sd[n] = torch.ones(sp.ref_shape) + 0.05 * torch.randn(sp.ref_shape, generator=g)
                else:
                    sd[n] = 0.02 * torch.randn(sp.ref_shape, generator=g)
            else:
                fan_in = math.prod(sp.ref_shape[1:])
                sd[n] = torch.randn(sp.ref_shape, generator=g) / math.sqrt(fan_in)
        self.load_state_dict(sd)
        return sd

    def load_state_dict(self, sd, strict=True):
        self.ps.load_state_dict(sd, strict)
        self.refresh_weights(cast_shadow=True)

    def state_dict(self):
        return self.ps.state_dict()

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def refresh_weights(self, cast_shadow=False, lazy=False):
        """bf16 operand copies derived from the f32 master: the fprop shadow (optionally; the fused
        AdamW kernel already refreshes it) and the transposed/flipped dgrad copies.  lazy: the copies only the BACKWARD pass reads
        (dgrad weights) are refreshed by the next forward() on the side stream, beside it (prep_side)."""
        ps = self.ps
        if cast_shadow and not self.f32:                 # (f32 mode: the operand copy IS the master)
            lib.call("siss_cast_f32_bf16", ps.flat, ps.shadow, ps.total)
        with lib.f32_mode(self.f32):
            for n, (wf, wd) in self._up_w.items():        # sub-pixel upsample sites: phase weights from the f32 master (f32 sums, one rounding)
                lib.call("siss_upsample_phase_weights", ps.p(n), wf, wd, wf.shape[2], wf.shape[3])
        if lazy and self.prep_side and self.wT and self.device.type == "cuda":
            self._wT_stale = True
            return
        self._refresh_dgrad_copies()

    def _refresh_dgrad_copies(self):
        ps = self.ps
        self._wT_stale = False
        if not self.wT:
            self._build_wt_jobs()
        # (from the bf16 shadow, which holds the rounded master at this point: cast above, or refreshed by the fused AdamW launch)
        with lib.f32_mode(self.f32):
            lib.call("siss_conv_weight_dgrad_multi_bf16", ps.shadow, self._wt_all, self._wt_jobs, self._wt_njobs,
                     self._wt_tiles)
        for pre, (buf, idx) in self._wds.items():
            torch.index_select(self.wT[pre + ".conv.weight"], 0, idx, out=buf)
        # conv_out dgrad operand: Wn^T, [Cin][K = 9*Cout padded to 64] bf16 (k = tap*Cout + co)
        w = ps.p("conv_out.weight")
        k = w.shape[0] * w.shape[1]
        if getattr(self, "_wd_out", None) is None:
            self._wd_out = torch.zeros(w.shape[2], -(-k // 64) * 64, dtype=self.adt, device=self.device)
        self._wd_out[:, :k] = w.reshape(k, w.shape[2]).t()

    def _ds_weights(self, pre, order):
        """Plane-grouped copy [9][Ci][Co] of a stride-2 conv's transposed weights (entry i = W[order[i]]^T); kept
        in step with the master by refresh_weights()."""
        ent = self._wds.get(pre)
        if ent is None:
            wT = self.wT[pre + ".conv.weight"]          # [9][Ci][Co], index 8 - tap holds W[tap]^T
            idx = torch.tensor([8 - tap for tap in order], device=self.device)
            ent = (torch.empty_like(wT), idx)
            torch.index_select(wT, 0, idx, out=ent[0])
            self._wds[pre] = ent
        return ent[0]

    def _build_wt_jobs(self):
        """One bf16 buffer holding every dgrad weight copy ([taps][Cin][Cout], tap order reversed) and the
        job table of the batched transpose kernel."""
        import numpy as np
        ps = self.ps
        jobs, off, tiles = [], 0, 0
        for n, sp in ps.specs.items():
            if sp.kind in ("conv3", "conv1", "mat") and n != "conv_out.weight" and "time_emb" not in n:
                t, co, ci = sp.native_shape if sp.kind == "conv3" else (1, *sp.native_shape)
                jobs.append((n, sp.off, off, t, co, ci, tiles))
                off += -(-t * co * ci // 64) * 64
                tiles += t * (-(-co // 64)) * (-(-ci // 64))        # 64 x 64 tiles: optimizer.hip conv_weight_dgrad_multi_kernel
        self._wt_all = torch.empty(off, dtype=self.adt, device=self.device)
        rec = np.zeros(len(jobs), dtype=np.dtype([("src", "<i8"), ("dst", "<i8"), ("taps", "<i4"), ("co", "<i4"),
                                                   ("ci", "<i4"), ("tile0", "<i4")]))
        for i, (n, src, dst, t, co, ci, t0) in enumerate(jobs):
            rec[i] = (src, dst, t, co, ci, t0)
            self.wT[n] = self._wt_all[dst:dst + t * co * ci].view(t, ci, co)
        self._wt_jobs = torch.from_numpy(rec.view(np.uint8)).to(self.device)
        self._wt_njobs, self._wt_tiles = len(jobs), tiles

    # ------------------------------------------------------------------ buffers
    def _act(self, name, n, h, w, c):
        k = (name, n, h, w, c)
        a = self._acts.get(k)
        if a is None:
            a = Act(n, h, w, c, self.device, dtype=self.adt)
            self._acts[k] = a
        return a

    def _buf(self, name, shape, dtype=torch.float32):
        k = (name, tuple(shape), dtype)
        b = self._bufs.get(k)
        if b is None:
            # slack: tile-granular kernels may READ (never use) up to one 256-B row past a small operand
            n = int(math.prod(shape))
            b = torch.zeros(n + 1024, dtype=dtype, device=self.device)[:n].view(shape)
            self._bufs[k] = b
        return b

    def _get(self, n, h, w, c):
        lst = self._pool.setdefault((n, h, w, c), [])
        a = lst.pop() if lst else Act(n, h, w, c, self.device, dtype=self.adt)
        self._wsync(a)
        return a

    def _wsync(self, a):
        """`a` is about to be overwritten: a queued (grouped) wgrad that still reads it must run first."""
        if a is None:
            return
        key = id(getattr(a, "base", a).buf)
        if self._held and key in self._held:
            self._flush_wgrads()
        if self._side_held and key in self._side_held:
            self._join_side()

    def _is_held(self, buf):
        return bool((self._held and id(buf) in self._held) or (self._side_held and id(buf) in self._side_held))

    def _join_side(self):
        """The launches on the side stream have to be complete before what follows on this stream; their operands return to the pool."""
        if not self._side_busy:
            return
        torch.cuda.current_stream().wait_stream(self._side)
        self._side_busy, self._side_held = False, {}
        rel, self._side_release = self._side_release, []
        for a in rel:
            self._put(a)

    def _flush_wgrads_side(self):
        """The queued weight-gradient products as CAPPED grouped launches on the side stream (behind everything issued so far)."""
        if not self._wq:
            return
        if not (self.wgrad_side and self.side_blocks >= 8) or self.f32:
            return self._flush_wgrads()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream())
        jobs = (lib.TNJob * len(self._wq))(*[j for j, _ in self._wq])
        with torch.cuda.stream(self._side):
            lib.call("siss_gemm_tn_grouped_capped", jobs, len(self._wq), int(self.side_blocks) & ~7)
            post, self._wq_post = self._wq_post, []
            for fn in post:                              # launches that consume a queued product (the sub-pixel upsample's tap fold)
                fn()
        for _, (dy, _x) in self._wq:                     # their cotangent operands stay out of the pool until the join
            buf = getattr(dy, "buf", None)
            if buf is not None and id(buf) in self._held:
                self._side_held[id(buf)] = self._held.pop(id(buf))
        keep = []
        for a in self._held_release:
            (self._side_release if id(a.buf) in self._side_held else keep).append(a)
        self._held_release = keep
        self._wq = []
        self._side_busy = True

    def _flush_wgrads(self):
        """Run the queued weight-gradient products as grouped launches and give their operands back to the pool."""
        if self._wq:
            jobs = (lib.TNJob * len(self._wq))(*[j for j, _ in self._wq])
            lib.call("siss_gemm_tn_grouped", jobs, len(self._wq))
            self._wq = []
        post, self._wq_post = self._wq_post, []
        for fn in post:                                  # launches that consume a queued product (the sub-pixel upsample's tap fold)
            fn()
        for job, _ in self._pair1:                      # one-panel top-resolution products that found no 3-tap partner: on their own
            self._launch_tn_job(job)
        self._pair1 = []
        self._held = {}
        rel, self._held_release = self._held_release, []
        for a in rel:
            self._put(a)

    def _ns_auto(self):
        return -2 if (self.wgrad_overwrite and not self.f32) else 0

    def _launch_tn_job(self, job):
        lib.call("siss_gemm_tn", job.Y, job.ldy, job.X, job.ldx, job.dW, job.set_stride, job.N, job.C, job.npanels,
                 lib.int_array(list(job.shifts)[:job.npanels]), lib.int_array(list(job.coffs)[:job.npanels]), job.nsets,
                 job.rows_per_set, job.x_set_rows, job.row_begin, job.row_end, 0, job.zero_page, job.dbias, job.dbias2)

    def _unhold(self, dy):
        """A queued product that read `dy` has been launched: give the buffer back if its owner already released it."""
        self._held.pop(id(dy.buf), None)
        for a in [a for a in self._held_release if a.buf is dy.buf]:
            self._held_release.remove(a)
            self._put(a)

    def _put(self, a):
        if a is not None:
            if self._held and id(a.buf) in self._held:   # still the operand of a queued wgrad: back to the pool after the flush
                self._held_release.append(a)
                return
            if self._side_held and id(a.buf) in self._side_held:     # ... or of one running on the side stream: after the join
                self._side_release.append(a)
                return
            self._pool.setdefault((a.n, a.h, a.w, a.c), []).append(a)

    def _name(self, base):
        self._uid += 1
        return f"{base}#{self._uid}"

    # gradient map (cotangent accumulation across multiple consumers, e.g. skip connections)
    def _give(self, act, g):
        cur = self.gmap.get(id(act))
        if cur is None:
            self.gmap[id(act)] = g
        else:
            self._wsync(cur)
            lib.call("siss_add_inplace", cur.data, g.data, cur.n, cur.h, cur.w, cur.c)
            self._put(g)

    def _take(self, act):
        return self.gmap.pop(id(act))

    # ------------------------------------------------------------------ primitive ops (fwd + taped bwd)
    def _gn_partial(self, n, h, w, c):
        words = lib.query("siss_gn_partial_words", n, h, w, c, self.cfg.norm_num_groups)
        assert words > 0
        return self._buf("gn_partial", (max(words, 1),))

    def _quad_stats_of(self, a, form=True):
        """The GroupNorm statistics entries of `a` in the persistent conv kernel's format: the ones its producer left, or (two-pass
        sites only) the ones siss_quad_stats forms from one read of it -- once per forward pass."""
        if a.qstats is not None:
            return a.qstats
        cache = getattr(self, "_qs_cache", None)          # (reset by _forward: an engine that does not, does not take this path)
        if (cache is None or not form or not self.quad_stats or self.f32 or a.h * a.w <= 1024 or a.c % 8 or a.c > 512
                or (a.h + 2) * (a.w + 2) < 256):
            return None
        key = (a.data.data_ptr(), a.c)
        qs = cache.get(key)
        if qs is None:
            qs = self._buf("quad_stats.%x.%d" % key, (lib.query("siss_conv_qstats_words", a.rows, a.c),))
            lib.call("siss_quad_stats", a.data, getattr(a, "ld", a.c), a.rows, a.c, a.rows_per_image, qs)
            self._qs_cache[key] = qs
        return qs

    def gn(self, x: Act, pre, silu, compact_out=False, eps=None):
        """GroupNorm(+SiLU).  Returns Act (or a compact [N*H*W, C] bf16 tensor)."""
        G, eps = self.cfg.norm_num_groups, (self.cfg.norm_eps if eps is None else eps)
        ps = self.ps
        nm = self._name(pre)
        mean = self._buf(nm + ".mean", (x.n, G))
        rstd = self._buf(nm + ".rstd", (x.n, G))
        if compact_out:
            y = self._buf(nm + ".y", (x.n * x.h * x.w, x.c), self.adt)
            yptr = y
        else:
            y = self._act(nm + ".y", x.n, x.h, x.w, x.c)
            yptr = y.data
        ldx = getattr(x, "ld", x.c)                 # x may be a column view of a concat buffer (ActView)
        # statistics left by the producing conv(s) (conv(want_stats=True)): x itself, or both parts of a concat
        qa = qb = None
        ca = x.c
        if self.epi_stats:
            parts = getattr(x, "cat_parts", None)
            # (quad_stats: a part whose producer left none -- conv_in, a downsample, a sub-pixel upsample -- gets them from one read
            #  of that part alone, kept for the forward pass: a skip tensor's second GroupNorm then has no statistics pass at all)
            form = (x.c // G) % 4 == 0 and x.c % G == 0   # (groups of whole 4-channel quads: what the finalize kernel folds)
            if parts is not None:
                sa = self._quad_stats_of(parts[0], form and parts[1].c <= 512 and parts[1].c % 8 == 0)
                sb = self._quad_stats_of(parts[1], form and sa is not None)
                if sa is not None and sb is not None:
                    qa, qb, ca = sa, sb, parts[0].c
            else:
                qa = self._quad_stats_of(x, form)
        if qa is not None:
            lib.call("siss_groupnorm_fwd_qs", x.data, ps.p(pre + ".weight"), ps.p(pre + ".bias"), yptr, mean, rstd,
                     self._gn_partial(x.n, x.h, x.w, x.c), qa, ca, qb, x.n, x.h, x.w, x.c, G, float(eps), int(silu),
                     int(compact_out), 0 if ldx == x.c else ldx)
        else:
            lib.call("siss_groupnorm_fwd_ld", x.data, ps.p(pre + ".weight"), ps.p(pre + ".bias"), yptr, mean, rstd,
                     self._gn_partial(x.n, x.h, x.w, x.c), x.n, x.h, x.w, x.c, G, float(eps), int(silu), int(compact_out),
                     0 if ldx == x.c else ldx)

        def bwd(dy, colsum=None, accum: Act = None, colsum_ld=0, accum2: Act = None, split=None):
            """dy: Act (padded) or compact tensor, nb samples.  Returns dx Act (nb samples).
            accum / accum2: cotangents already known for x (added; accum is overwritten in place).
            split = (da, db, accumulate_b): x was concat(a, b) -- write the two halves straight into da / db."""
            nb = self.nb
            # accum is normally overwritten in place; not while a queued (grouped) wgrad still reads it: fresh output then
            held = accum is not None and self._is_held(accum.buf)
            if accum is not None and not held:
                self._wsync(accum)                      # written in place
            s2d = False
            if split is not None:
                da, db, accb = split[:3]
                s2d = len(split) > 3 and bool(split[3])   # da is a half-resolution, 4x-channel Act: the first part in space-to-depth layout
                dx, dx2p, split_c = da, db.data, (da.c // 4 if s2d else da.c)
                self._wsync(db)
            else:
                dx = accum if (accum is not None and not held) else self._get(nb, x.h, x.w, x.c)
                dx2p, split_c, accb = None, 0, False
            dyp = dy.data if isinstance(dy, Act) else dy
            lib.call("siss_groupnorm_bwd_ld_s2d" if s2d else "siss_groupnorm_bwd_ld", dyp, x.data, ps.p(pre + ".weight"), ps.p(pre + ".bias"), mean, rstd,
                     dx.data, accum.data if accum is not None else None,
                     accum2.data if accum2 is not None else None, dx2p, split_c, int(accb),
                     ps.g(pre + ".weight", self.gbase), ps.g(pre + ".bias", self.gbase), colsum, colsum_ld,
                     self._gn_partial(nb, x.h, x.w, x.c), nb, x.n, self.set_images, ps.total,
                     x.h, x.w, x.c, G, int(silu), int(not isinstance(dy, Act)), 0 if ldx == x.c else ldx)
            if held and split is None:
                self._put(accum)                        # read only here: released once the queued wgrad has run
            return dx
        return y, bwd

    def conv(self, x: Act, pre, ksize=3, rowbias=None, residual: Act = None, out_name=None, ldrb=None, cat_with=None,
             skip_head=None, want_stats=False, shortcut=None, fprop=True):
        """stride-1 'same' conv (3x3 or 1x1) with fused bias / time-embedding row bias / residual.  cat_with: the skip
        activation the result is about to be concatenated with -- the result is then written straight into the head
        columns of that concat buffer (epilogue with ldc = C + C_skip) and concat() only copies the skip.
        shortcut = (x2, pre2): also add the 1x1 convolution `pre2` of x2 (a resnet's conv_shortcut) -- folded into the 3x3
        product when its kernel takes it (ops.conv_fprop_sc), otherwise as a product of its own whose result is the residual.
        fprop=False: no forward launch, only the backward closure (the folded shortcut's own dgrad / wgrad)."""
        ps = self.ps
        w = ps.sh(pre + ".weight")
        if ksize == 1:
            w = w.view(1, *w.shape)
        co = w.shape[1]
        if not fprop:
            y = None
        elif cat_with is not None and self.direct_cat:
            assert (cat_with.n, cat_with.h, cat_with.w) == (x.n, x.h, x.w)
            if isinstance(cat_with, ActView):       # the skip already lives in the tail columns of its concat buffer
                assert cat_with.c0 == co and cat_with.base.c == co + cat_with.c
                y = ActView(cat_with.base, 0, co)
            else:
                y = ActView(self._act(self._name("cat"), x.n, x.h, x.w, co + cat_with.c), 0, co)
        elif skip_head is not None and self.direct_cat:
            # this result is a SKIP: it goes straight into the tail columns of the concat buffer it will meet its
            # up-path partner (skip_head channels) in; every down-path reader takes the row stride (ld)
            y = ActView(self._act(self._name("cat"), x.n, x.h, x.w, skip_head + co), skip_head, co)
        else:
            y = self._act(self._name(out_name or pre), x.n, x.h, x.w, co)
        # want_stats: the result feeds a GroupNorm -- at the sites where that GroupNorm would make a statistics pass of its
        # own (more than 32 x 32 pixels) the conv's epilogue leaves the statistics (y.qstats) when its kernel can
        if fprop:
            y.qstats = None
            stats = want_stats and self.epi_stats and ksize == 3 and x.h * x.w > 1024 and co % 128 == 0
            qs = self._buf(self._name(pre) + ".qs", (lib.query("siss_conv_qstats_words", x.rows, co),)) if stats else None
            folded = False
            if shortcut is not None:
                x2, pre2 = shortcut
                assert residual is None and ksize == 3
                if self.fold_shortcut and ops.conv3x3_sc_takes(x, co, y, x2):
                    if ops.conv_fprop_sc(x, w, y, x2, ps.sh(pre2 + ".weight"), bias=ps.p(pre + ".bias"), bias2=ps.p(pre2 + ".bias"),
                                         rowbias=rowbias, ldrb=ldrb, qstats=qs):
                        y.qstats = qs
                    folded = True
                else:
                    residual, _ = self.conv(x2, pre2, ksize=1)
            if folded:
                pass
            elif stats:
                if ops.conv_fprop_qstats(x, w, y, qs, bias=ps.p(pre + ".bias"), rowbias=rowbias, residual=residual, ldrb=ldrb):
                    y.qstats = qs
            else:
                ops.conv_fprop(x, w, y, bias=ps.p(pre + ".bias"), rowbias=rowbias, residual=residual, ksize=ksize, ldrb=ldrb)

        def bwd(dy: Act, need_dx=True, accum: Act = None, bias_grad=True, bias_grad2=None, sc_dgrad=None, wgrad=True):
            """sc_dgrad = (pre2, c2, get_out2): also form the dgrad of the 1x1 convolution `pre2` (c2 input channels) over the SAME
            cotangent into get_out2() -- inside this 3x3 dgrad when its kernel takes it (returns (dx, out2)), else not at all (returns
            (dx, None): the caller runs it, and no buffer was taken).  wgrad=False: the weight gradient has been taken already."""
            if wgrad:
                dW = ps.grads[self.gbase:, ps.specs[pre + ".weight"].off:]
                # the bias gradient (column sums of dy) rides along in the wgrad GEMM as one more product
                self._wgrad(dy, x, dW, co, x.c, ksize, dbias=ps.g(pre + ".bias", self.gbase) if bias_grad else None,
                            dbias2=bias_grad2)
            if not need_dx:
                return None
            if accum is not None:
                self._wsync(accum)
            dx = accum if accum is not None else self._get(dy.n, x.h, x.w, x.c)
            if sc_dgrad is not None:
                pre2, c2, get_out2 = sc_dgrad
                # (not with more than twice the 3x3 product's columns: every x tile pays a tile's fixed epilogue for a third of its
                # MFMAs -- measured at 128 x 128, 128 -> 128 with a 384-channel shortcut: 424 us folded against 129 + 149 us apart)
                # The x-shaped target (the concat-wide input on the up path) is only taken from the pool once the fold is decided.
                if (self.fold_shortcut and ksize == 3 and c2 <= 2 * x.c
                        and ops.conv3x3_dgrad_sc_takes(dy, x.c, dx, SimpleNamespace(c=c2, ld=c2), residual=accum)):
                    out2 = get_out2()
                    ops.conv_dgrad_sc(dy, self.wT[pre + ".weight"], dx, self.wT[pre2 + ".weight"][0], out2, residual=accum)
                    return dx, out2
                ops.conv_dgrad(dy, self.wT[pre + ".weight"], dx, residual=accum, ksize=ksize)
                return dx, None
            ops.conv_dgrad(dy, self.wT[pre + ".weight"], dx, residual=accum, ksize=ksize)
            return dx
        return y, bwd

    def _wgrad(self, dy: Act, x: Act, dW_view, co, ci, ksize, shifts=None, coffs=None, ldx=None, dbias=None,
               dbias2=None):
        """dW_view: grads[gbase:, off:] -- a strided view whose [0,0] element is the target."""
        ps = self.ps
        if shifts is None:
            if ksize == 3:
                from .layout import conv3x3_panels
                shifts, coffs = conv3x3_panels(dy.wp, ci)
            else:
                shifts, coffs = [0], [0]
        t = len(shifts)
        rows_per_set = self.set_images * dy.rows_per_image
        assert x.n in (dy.n, self.set_images)
        x_set_rows = rows_per_set if x.n == dy.n else 0      # 0: every set reads the same saved rows
        rb, re = dy.wp + 1, rows_per_set - (dy.wp + 1)
        tiles = (-(-co // 128)) * (-(-ci // 128))
        ns = ops._nsplits(tiles, t, self.nsets, re - rb, ops.is_conv3_panels(shifts, coffs))
        sh, cf, zp = lib.int_array(shifts), lib.int_array(coffs), ops.zero_page(self.device)
        nsets = self.nsets
        if self.group_rows and re - rb <= self.group_rows and isinstance(dy, Act):
            job = lib.TNJob(Y=dy.data.data_ptr(), ldy=dy.c, X=x.data.data_ptr(), ldx=ldx or getattr(x, "ld", x.c),
                            dW=dW_view.data_ptr(), set_stride=ps.total, N=co, C=ci, npanels=t, nsets=nsets,
                            rows_per_set=rows_per_set, row_begin=rb, row_end=re, nsplits=self._ns_auto(), x_set_rows=x_set_rows,
                            zero_page=zp.data_ptr(), dbias=dbias.data_ptr() if dbias is not None else None,
                            dbias2=dbias2.data_ptr() if dbias2 is not None else None,
                            shifts=(lib.I * 9)(*shifts, *([0] * (9 - t))), coffs=(lib.I * 9)(*coffs, *([0] * (9 - t))))
            self._wq.append((job, (dy, x)))
            self._held[id(dy.buf)] = dy
            if len(self._wq) >= self.group_max:
                self._flush_wgrads_side() if (self._side_phase and self.side_follow) else self._flush_wgrads()
            return
        if (self.pair_top and not self.f32 and isinstance(dy, Act) and re - rb >= self.pair_min_rows
                and (t == 1 or (ops.is_conv3_panels(shifts, coffs) and self._pair1))):
            job = lib.TNJob(Y=dy.data.data_ptr(), ldy=dy.c, X=x.data.data_ptr(), ldx=ldx or getattr(x, "ld", x.c),
                            dW=dW_view.data_ptr(), set_stride=ps.total, N=co, C=ci, npanels=t, nsets=nsets,
                            rows_per_set=rows_per_set, row_begin=rb, row_end=re, nsplits=self._ns_auto(), x_set_rows=x_set_rows,
                            zero_page=zp.data_ptr(), dbias=dbias.data_ptr() if dbias is not None else None,
                            dbias2=dbias2.data_ptr() if dbias2 is not None else None,
                            shifts=(lib.I * 9)(*shifts, *([0] * (9 - t))), coffs=(lib.I * 9)(*coffs, *([0] * (9 - t))))
            if t == 1:                                  # waits for the next 3-tap product (or the end of the backward pass)
                self._pair1.append((job, dy))
                self._held[id(dy.buf)] = dy
                return
            # the partner: a waiting product over the SAME cotangent (the resnet's own conv_shortcut: its Y tiles are in L2), else the oldest
            same = [i for i, (_, d) in enumerate(self._pair1) if d is dy]
            j1, dy1 = self._pair1.pop(same[0] if same else 0)
            # (the launcher refuses a pair it cannot split over the device -- fewer CUs than the 3-tap product's base grid needs, no
            # room for a one-tap block: status 1 -- the two products then run as launches of their own)
            if lib.call("siss_gemm_tn_pair", lib.C.byref(job), lib.C.byref(j1), 0, refusable=True):
                self._launch_tn_job(job)
                self._launch_tn_job(j1)
            if not any(d is dy1 for _, d in self._pair1):
                self._unhold(dy1)
            return
        lib.call("siss_gemm_tn", dy.data, dy.c, x.data, ldx or getattr(x, "ld", x.c), dW_view, ps.total, co, ci, t,
                 sh, cf, nsets, rows_per_set, x_set_rows, rb, re, ns, zp, dbias, dbias2)

    # ------------------------------------------------------------------ time embedding
    def time_embed(self, t):
        cfg, ps = self.cfg, self.ps
        B = t.shape[0]
        c0 = cfg.block_out_channels[0]
        e0 = self._buf("temb.e0", (B, c0))
        h1 = self._buf("temb.h1", (B, self.temb_dim))
        emb = self._buf("temb.emb", (B, self.temb_dim))
        lib.call("siss_timestep_sincos", t, e0, B, c0, int(cfg.flip_sin_to_cos), float(cfg.freq_shift))
        lib.call("siss_linear_small_fwd", e0, ps.p("time_embedding.linear_1.weight"),
                 ps.p("time_embedding.linear_1.bias"), h1, B, self.temb_dim, c0, 0)
        lib.call("siss_linear_small_fwd", h1, ps.p("time_embedding.linear_2.weight"),
                 ps.p("time_embedding.linear_2.bias"), emb, B, self.temb_dim, self.temb_dim, 1)
        self.emb = emb
        # every resnet's time_emb_proj(silu(emb)) in ONE launch: tp_all[:, c0:c0+cout] is its row bias
        self.tp_all = self._buf("temb.tp_all", (B, self.temb_ntot))
        lib.call("siss_linear_multi_fwd", emb, ps.flat, self.temb_woff, self.temb_boff, self.tp_all, B,
                 self.temb_ntot, self.temb_dim)

        def bwd():
            nb, T = self.nb, self.temb_dim
            d_s = self._buf("temb.d_s", (nb, T))           # grad wrt silu(emb)
            d_s1 = self._buf("temb.d_s1", (nb, T))
            gb = self.gbase
            # all time_emb_proj weight/bias grads (+ the conv1 biases that share them) and d_s, batched
            lib.call("siss_linear_multi_bwd", self.dtp_all, emb, ps.flat, ps.grads[gb:], self.temb_woff,
                     self.temb_boff, self.temb_boff2, d_s, nb, B, self.set_images, ps.total, self.temb_ntot, T)
            lib.call("siss_linear_small_bwd", d_s, emb, h1, ps.p("time_embedding.linear_2.weight"), d_s1, 0,
                     ps.g("time_embedding.linear_2.weight", gb), ps.g("time_embedding.linear_2.bias", gb), None,
                     nb, B, self.set_images, ps.total, ps.total, T, T, 1)
            lib.call("siss_linear_small_bwd", d_s1, h1, e0, ps.p("time_embedding.linear_1.weight"), None, 0,
                     ps.g("time_embedding.linear_1.weight", gb), ps.g("time_embedding.linear_1.bias", gb), None,
                     nb, B, self.set_images, ps.total, ps.total, T, c0, 0)
        self.tape.append(bwd)

    # ------------------------------------------------------------------ blocks
    def resnet(self, x: Act, pre, cat_with=None, skip_head=None):
        ps = self.ps
        cin = x.c
        cout = ps.specs[pre + ".conv1.weight"].ref_shape[0]
        B = x.n
        a1, gn1_b = self.gn(x, pre + ".norm1", True)
        col0, _ = self.temb_cols[pre]
        h, c1_b = self.conv(a1, pre + ".conv1", rowbias=self.tp_all[:, col0:], ldrb=self.temb_ntot, want_stats=True)
        a2, gn2_b = self.gn(h, pre + ".norm2", True)
        has_sc = cin != cout
        if has_sc:
            # conv_shortcut(x) + conv2(a2) as one accumulation where the 3x3 kernel takes it (else: a 1x1 product + residual);
            # the shortcut's backward (dgrad into x's cotangent, wgrad) stays a product of its own
            _, sc_b = self.conv(x, pre + ".conv_shortcut", ksize=1, fprop=False)
            out, c2_b = self.conv(a2, pre + ".conv2", cat_with=cat_with, skip_head=skip_head, want_stats=True,
                                  shortcut=(x, pre + ".conv_shortcut"))
        else:
            out, c2_b = self.conv(a2, pre + ".conv2", residual=x, cat_with=cat_with, skip_head=skip_head, want_stats=True)

        def bwd():
            nb = self.nb
            dout = self._take(out)
            gb = self.gbase
            # conv2 (its bias gradient equals the shortcut conv's bias gradient: same pre-activation).  With a shortcut, its dgrad
            # (dout . W_sc, HBM-bound on its own) rides in conv2's 3x3 dgrad over the same cotangent where that kernel takes it
            acc_sc = None
            if has_sc:
                # the shortcut's weight gradient first: it reduces over the same cotangent as conv2's, and at the top resolution it is
                # queued for the NEXT 3-tap weight-gradient launch -- conv2's, two lines down (pair_top)
                sc_b(dout, bias_grad=False, need_dx=False)
                da2, acc_sc = c2_b(dout, bias_grad2=ps.g(pre + ".conv_shortcut.bias", gb),
                                   sc_dgrad=(pre + ".conv_shortcut", x.c, lambda: self._get(nb, x.h, x.w, x.c)))
            else:
                da2 = c2_b(dout)
            # column sums of dh = cotangent of time_emb_proj's output (and of conv1's bias); the weight
            # gradients of ALL time_emb_proj layers are formed in one batched launch at the end
            dh = gn2_b(da2, colsum=self.dtp_all[:, col0:], accum=None, colsum_ld=self.temb_ntot)
            self._put(da2)
            da1 = c1_b(dh, bias_grad=False)
            self._put(dh)
            prior = self.gmap.pop(id(x), None)          # cotangent x already received from another consumer
            if has_sc:
                acc = acc_sc if acc_sc is not None else sc_b(dout, bias_grad=False, wgrad=False)    # (dgrad only: see above)
                self._put(dout)
            else:
                acc = dout
            parts = getattr(x, "cat_parts", None)
            if parts is not None:
                # x = concat(a, b): norm1's backward writes d_a and d_b (+= the skip's running cotangent) directly
                a, b = parts
                # (a sub-pixel upsample's output takes its cotangent in space-to-depth layout straight from this kernel's store path)
                s2d = bool(getattr(a, "s2d_cot", False)) and self.gmap.get(id(a)) is None
                da = self._get(nb, a.h // 2, a.w // 2, 4 * a.c) if s2d else self._get(nb, a.h, a.w, a.c)
                accb = self.gmap.get(id(b))
                db = accb if accb is not None else self._get(nb, b.h, b.w, b.c)
                gn1_b(da1, accum=acc, accum2=prior, split=(da, db, accb is not None, s2d))
                self.gmap[id(b)] = db
                self._put(acc)
                self._give(a, da)
                x.cat_done = True
            else:
                dx = gn1_b(da1, accum=acc, accum2=prior)
                self.gmap[id(x)] = dx
            self._put(prior)
            self._put(da1)
        self.tape.append(bwd)
        return out

    def attention(self, x: Act, pre):
        """Spatial self-attention block.  attention_head_dim = C (CelebA-HQ: one 512-wide head) runs QK^T / PV
        on the MFMA GEMMs; small head dims (diffusers default 8: the MNIST UNet) use the per-(sample, head)
        online-softmax kernels."""
        ps = self.ps
        C, B, S = x.c, x.n, x.h * x.w
        D = self.cfg.head_dim(C)
        if (self.fused_attn and not self.f32 and D == C and x.w % 4 == 0 and not isinstance(x, ActView)
                and lib.query("siss_attn1h_takes", S, C)):
            return self._attention_fused(x, pre)
        small = D != C
        assert not small or D in (8, 16, 32), f"attention_head_dim {D} is not covered by the HIP kernels"
        scale = D ** -0.5
        rows = B * S
        nm = self._name(pre)
        hn, gn_b = self.gn(x, pre + ".group_norm", False, compact_out=True)
        bb = lambda s, shape: self._buf(nm + s, shape, self.adt)
        q, k, v = bb(".q", (rows, C)), bb(".k", (rows, C)), bb(".v", (rows, C))
        vT, sc, p = bb(".vT", (B, C, S)), bb(".s", (B, S, S)), bb(".p", (B, S, S))
        o, y = bb(".o", (rows, C)), bb(".y", (rows, C))

        def lin(a, wname, out, n_rows, bias=True):
            ops.gemm_nt(lib.ptr(a), C, ps.sh(wname + ".weight"), lib.ptr(out), C, n_rows, C, C, [0], [0],
                        bias=ps.p(wname + ".bias") if bias else None)
        lin(hn, pre + ".to_q", q, rows); lin(hn, pre + ".to_k", k, rows); lin(hn, pre + ".to_v", v, rows)
        if small:
            lse = self._buf(nm + ".lse", (B, C // D, S))
            lib.call("siss_mha_small_fwd", q, k, v, o, lse, B, S, C, D, float(scale))
        else:
            lib.call("siss_transpose_bf16", v, vT, B, S, C)
            ops.gemm_nt(lib.ptr(q), C, k, lib.ptr(sc), S, S, S, C, [0], [0], alpha=scale, batch=B,
                        stride_a=S * C, stride_w=S * C, stride_c=S * S)
            if S <= 1024:
                lib.call("siss_softmax_fwd", sc, p, B * S, S)
            else:                                   # long rows (the VAE's 64x64 mid attention): whole-row kernel
                lib.call("siss_softmax_rows_fwd", sc, p, B * S, S, S, 0)
            ops.gemm_nt(lib.ptr(p), S, vT, lib.ptr(o), C, S, C, S, [0], [0], batch=B,
                        stride_a=S * S, stride_w=C * S, stride_c=S * C)
        lin(o, pre + ".to_out.0", y, rows)
        out = self._act(nm + ".out", B, x.h, x.w, C)
        lib.call("siss_compact_add_to_pad", y, x.data, out.data, B, x.h, x.w, C)

        def bwd():
            nb, gb, ns, si = self.nb, self.gbase, self.nsets, self.set_images
            rows2 = nb * S
            dout = self._take(out)
            tb = lambda s, shape, dt=self.adt: self._buf("attn" + s, shape, dt)
            # the operands of this site's four weight-gradient products get buffers of their OWN (8 MB each at B = 16): the
            # products are queued and run later in a grouped launch, when the shared scratch has long been reused
            own = (lambda s, shape: self._buf(nm + ".bwd" + s, shape, self.adt)) if self.group_attn and self.group_rows else tb
            dy = own(".dy", (rows2, C))
            lib.call("siss_pad_to_compact", dout.data, dy, nb, x.h, x.w, C)
            zp = ops.zero_page(self.device)

            def lin_bwd(dyt, xin, wname, dx_out, accumulate):
                """dyt [rows2,C] cotangent of y = xin W^T + b (xin has B*S rows shared by the sets)."""
                dW = ps.grads[gb:, ps.specs[wname + ".weight"].off:]
                tiles = (-(-C // 128)) ** 2
                if self.group_attn and self.group_rows:
                    z9 = (lib.I * 9)(*([0] * 9))
                    self._wq.append((lib.TNJob(Y=dyt.data_ptr(), ldy=C, X=xin.data_ptr(), ldx=C, dW=dW.data_ptr(),
                                               set_stride=ps.total, N=C, C=C, npanels=1, nsets=ns, rows_per_set=si * S,
                                               row_begin=0, row_end=si * S, nsplits=self._ns_auto(),
                                               x_set_rows=si * S if B == nb else 0, zero_page=zp.data_ptr(),
                                               dbias=ps.g(wname + ".bias", gb).data_ptr(), dbias2=None, shifts=z9, coffs=z9),
                                     (dyt, xin)))
                else:
                    lib.call("siss_gemm_tn", dyt, C, xin, C, dW, ps.total, C, C, 1, lib.int_array([0]),
                             lib.int_array([0]), ns, si * S, si * S if B == nb else 0,
                             0, si * S, ops._nsplits(tiles, 1, ns, si * S, False), zp, ps.g(wname + ".bias", gb), None)
                if dx_out is not None:
                    ops.gemm_nt(lib.ptr(dyt), C, self.wT[wname + ".weight"], lib.ptr(dx_out), C, rows2, C, C,
                                [0], [0], res_ptr=lib.ptr(dx_out) if accumulate else None, ldr=C)
            do = tb(".do", (rows2, C))
            lin_bwd(dy, o, pre + ".to_out.0", do, False)
            dqkv = own(".dqkv", (3, rows2, C))        # one buffer: the three projections' dgrads run as ONE three-panel product
            dq, dk, dv = dqkv[0], dqkv[1], dqkv[2]
            if small:
                lib.call("siss_mha_small_bwd", q, k, v, o, lse, do, dq, dk, dv, nb, B, S, C, D, float(scale))
            else:
                dp, ds = tb(".dp", (nb, S, S)), tb(".ds", (nb, S, S))
                dkf, dvf = tb(".dkf", (nb, S, C), torch.float32), tb(".dvf", (nb, S, C), torch.float32)
                kT = tb(".kT", (B, C, S))
                lib.call("siss_transpose_bf16", k, kT, B, S, C)
                for g in range(nb // B):      # cotangent groups that share the B forward samples
                    sl = slice(g * B, (g + 1) * B)
                    # dP = dO V^T
                    ops.gemm_nt(lib.ptr(do[g * B * S:]), C, v, lib.ptr(dp[sl]), S, S, S, C, [0], [0], batch=B,
                                stride_a=S * C, stride_w=S * C, stride_c=S * S)
                    # dV[key][c] = sum_q P[q][key] dO[q][c]
                    lib.call("siss_gemm_tn", p, S, do[g * B * S:], C, dvf[sl], S * C, S, C, 1, lib.int_array([0]),
                             lib.int_array([0]), B, S, S, 0, S, -1, zp, None, None)   # -1: overwrite, no zero fill
                if S <= 1024:
                    lib.call("siss_softmax_bwd", p, dp, ds, nb * S, B * S, S, float(scale))
                else:
                    lib.call("siss_softmax_rows_bwd", p, dp, ds, nb * S, B * S, S, S, float(scale))
                for g in range(nb // B):
                    sl = slice(g * B, (g + 1) * B)
                    # dQ = dS K
                    ops.gemm_nt(lib.ptr(ds[sl]), S, kT, lib.ptr(dq[g * B * S:]), C, S, C, S, [0], [0], batch=B,
                                stride_a=S * S, stride_w=C * S, stride_c=S * C)
                    # dK[key][c] = sum_q dS[q][key] Q[q][c]
                    lib.call("siss_gemm_tn", ds[sl], S, q, C, dkf[sl], S * C, S, C, 1, lib.int_array([0]),
                             lib.int_array([0]), B, S, S, 0, S, -1, zp, None, None)   # -1: overwrite, no zero fill
                lib.call("siss_cast_f32_bf16", dkf, dk, dkf.numel())
                lib.call("siss_cast_f32_bf16", dvf, dv, dvf.numel())
            dhn = tb(".dhn", (rows2, C))
            lin_bwd(dq, hn, pre + ".to_q", None, False)
            lin_bwd(dk, hn, pre + ".to_k", None, False)
            lin_bwd(dv, hn, pre + ".to_v", None, False)
            # dhn = dq Wq + dk Wk + dv Wv: three panels of one product -- panel p reads rows [p * rows2, (p + 1) * rows2) of the
            # stacked cotangents (row shift) against the p-th of the three consecutive transposed weight copies
            wq, wk, wv = (self.wT[pre + n + ".weight"] for n in (".to_q", ".to_k", ".to_v"))
            esz = wq.element_size()
            assert wk.data_ptr() == wq.data_ptr() + esz * C * C and wv.data_ptr() == wk.data_ptr() + esz * C * C
            ops.gemm_nt(lib.ptr(dqkv), C, wq, lib.ptr(dhn), C, rows2, C, C, [0, rows2, 2 * rows2], [0, 0, 0])
            dx = gn_b(dhn, accum=dout)        # residual path: d_out passes straight through
            self._give(x, dx)
        self.tape.append(bwd)
        return out

    def _attention_fused(self, x: Act, pre):
        """The single-head attention block on the fused kernels (csrc/attn1h.hip):
            GroupNorm (compact rows) -> ONE q / k / v projection [rows, 3C] -> siss_attn1h_fwd (o in the padded layout, LSE kept)
            -> to_out as a 1x1 convolution whose epilogue adds the residual x;
        backward: to_out's dgrad (+ its queued wgrad) -> siss_attn1h_bwd (dq | dk | dv into ONE [rows2, 3C] buffer) -> one wgrad job
        (N = 3C) + one three-panel dgrad -> GroupNorm backward with the residual cotangent.  No S x S matrix, transpose, pad <->
        compact copy or f32 dK / dV scratch exists."""
        ps = self.ps
        C, B, S = x.c, x.n, x.h * x.w
        scale = float(C) ** -0.5
        rows = B * S
        nm = self._name(pre)
        hn, gn_b = self.gn(x, pre + ".group_norm", False, compact_out=True)
        sq, sk, sv = (ps.specs[pre + n] for n in (".to_q.weight", ".to_k.weight", ".to_v.weight"))
        bq, bk, bv = (ps.specs[pre + n] for n in (".to_q.bias", ".to_k.bias", ".to_v.bias"))
        assert sk.off == sq.off + C * C and sv.off == sk.off + C * C and bk.off == bq.off + C and bv.off == bk.off + C
        qkv = self._buf(nm + ".qkv", (rows, 3 * C), self.adt)
        lse = self._buf(nm + ".lse", (rows,))
        ops.gemm_nt(lib.ptr(hn), C, ps.sh(pre + ".to_q.weight"), lib.ptr(qkv), 3 * C, rows, 3 * C, C, [0], [0],
                    bias=ps.p(pre + ".to_q.bias"))
        o = self._act(nm + ".o", B, x.h, x.w, C)         # halo rows are never written: they keep their zeros
        lib.call("siss_attn1h_fwd", qkv, qkv[:, C:], qkv[:, 2 * C:], 3 * C, o.data, C, x.w, lse, B, S, C, scale)
        out, out_b = self.conv(o, pre + ".to_out.0", ksize=1, residual=x)

        def bwd():
            nb, gb, ns, si = self.nb, self.gbase, self.nsets, self.set_images
            rows2 = nb * S
            dout = self._take(out)
            do = out_b(dout)                            # to_out: weight / bias gradient (queued) and dgrad
            grouped = bool(self.group_attn and self.group_rows)
            # (a queued wgrad reads its cotangent operand long after this closure has returned: a buffer of the site's own then)
            dqkv = self._buf((nm + ".bwd" if grouped else "attn") + ".dqkv", (rows2, 3 * C), self.adt)
            delta = self._buf("attn.delta", (rows2,))
            lib.call("siss_attn1h_bwd", qkv, qkv[:, C:], qkv[:, 2 * C:], 3 * C, o.data, C, do.data, C, x.w, lse, delta,
                     dqkv, dqkv[:, C:], dqkv[:, 2 * C:], 3 * C, nb, B, S, C, scale)
            self._put(do)
            # [dWq ; dWk ; dWv] = dqkv^T hn and the three bias gradients: ONE product with N = 3C
            dW = ps.grads[gb:, sq.off:]
            zp = ops.zero_page(self.device)
            xsr = si * S if B == nb else 0
            if grouped:
                z9 = (lib.I * 9)(*([0] * 9))
                self._wq.append((lib.TNJob(Y=dqkv.data_ptr(), ldy=3 * C, X=hn.data_ptr(), ldx=C, dW=dW.data_ptr(),
                                           set_stride=ps.total, N=3 * C, C=C, npanels=1, nsets=ns, rows_per_set=si * S,
                                           row_begin=0, row_end=si * S, nsplits=self._ns_auto(), x_set_rows=xsr, zero_page=zp.data_ptr(),
                                           dbias=ps.g(pre + ".to_q.bias", gb).data_ptr(), dbias2=None, shifts=z9, coffs=z9),
                                 (dqkv, hn)))
                if len(self._wq) >= self.group_max:
                    self._flush_wgrads_side() if (self._side_phase and self.side_follow) else self._flush_wgrads()
            else:
                lib.call("siss_gemm_tn", dqkv, 3 * C, hn, C, dW, ps.total, 3 * C, C, 1, lib.int_array([0]), lib.int_array([0]),
                         ns, si * S, xsr, 0, si * S, 0, zp, ps.g(pre + ".to_q.bias", gb), None)
            # dhn = dq Wq + dk Wk + dv Wv: three panels (column windows of dqkv) against the three consecutive transposed copies
            wq, wk, wv = (self.wT[pre + n + ".weight"] for n in (".to_q", ".to_k", ".to_v"))
            esz = wq.element_size()
            assert wk.data_ptr() == wq.data_ptr() + esz * C * C and wv.data_ptr() == wk.data_ptr() + esz * C * C
            dhn = self._buf("attn.dhn", (rows2, C), self.adt)
            ops.gemm_nt(lib.ptr(dqkv), 3 * C, wq, lib.ptr(dhn), C, rows2, C, C, [0, 0, 0], [0, C, 2 * C])
            dx = gn_b(dhn, accum=dout)                  # residual path: d_out passes straight through
            self._give(x, dx)
        self.tape.append(bwd)
        return out

    def downsample(self, x: Act, pre, skip_head=None):
        """3x3 stride-2 conv (Downsample2D) as nine row-shifted panels over a space-to-depth copy."""
        ps, cfg = self.ps, self.cfg
        C, B, Ho, Wo = x.c, x.n, x.h // 2, x.w // 2
        z = self._act(self._name(pre + ".z"), B, Ho, Wo, 4 * C)
        ldx = getattr(x, "ld", C)
        lib.call("siss_space_to_depth_ld", x.data, z.data, B, x.h, x.w, C, 0 if ldx == C else ldx)
        wp = Wo + 2
        shifts, coffs = [], []
        for ky in range(3):
            for kx in range(3):
                if cfg.downsample_padding == 0:      # F.pad (0,1,0,1) then stride-2, pad 0
                    dy_, py, dx_, px = ky >> 1, ky & 1, kx >> 1, kx & 1
                else:                                # stride-2, pad 1
                    dy_, py, dx_, px = (ky - 1) >> 1, (ky - 1) & 1, (kx - 1) >> 1, (kx - 1) & 1
                shifts.append(dy_ * wp + dx_)
                coffs.append((py * 2 + px) * C)
        w = ps.sh(pre + ".conv.weight")
        if skip_head is not None and self.direct_cat:
            y = ActView(self._act(self._name("cat"), B, Ho, Wo, skip_head + C), skip_head, C)
        else:
            y = self._act(self._name(pre + ".y"), B, Ho, Wo, C)
        ops.gemm_nt(lib.ptr(z.data), 4 * C, w, lib.ptr(y.data), getattr(y, "ld", C), z.rows, C, C, shifts, coffs,
                    bias=ps.p(pre + ".conv.bias"), rows_per_image=z.rows_per_image, hp=z.hp, wp=z.wp)
        # dgrad: the taps that read the same space-to-depth plane (py, px) WRITE the same plane of dz, so each
        # plane is one multi-panel GEMM (4 / 2 / 2 / 1 taps) over a plane-grouped copy of the transposed weights
        planes = {}
        for tap in range(9):
            planes.setdefault(coffs[tap] // C, []).append(tap)
        order = [tap for pl in sorted(planes) for tap in planes[pl]]
        wds = None if getattr(self, "forward_only", False) else self._ds_weights(pre, order)

        def bwd():
            nb, gb = self.nb, self.gbase
            dy = self._take(y)
            dW = ps.grads[gb:, ps.specs[pre + ".conv.weight"].off:]
            self._wgrad(dy, z, dW, C, C, 3, shifts=shifts, coffs=coffs, ldx=4 * C, dbias=ps.g(pre + ".conv.bias", gb))
            acc = self.gmap.get(id(x))
            self._wsync(acc)
            dx = acc if acc is not None else self._get(nb, x.h, x.w, C)
            pos = 0
            if self.d2s_epilogue:
                # each plane GEMM writes its pixels straight to their place in dx (and adds the cotangent x already has):
                # no dz tensor, no depth-to-space pass
                if self.phase_launch and (not self.f32 or self.f32_fused) and len(planes) == 4:
                    p0 = [0]                             # the four planes' products as ONE launch
                    for plane in sorted(planes):
                        p0.append(p0[-1] + len(planes[plane]))
                    lib.call("siss_gemm_nt_d2s_phases", dy.data, C, wds, dx.data, C, None, dx.data if acc is not None else None, C,
                             dy.rows, C, C, lib.int_array(p0), lib.int_array([-shifts[tap] for tap in order]),
                             lib.int_array([0] * 9), dy.rows_per_image, dy.hp, dy.wp)
                else:
                    for plane in sorted(planes):
                        taps = planes[plane]
                        lib.call("siss_gemm_nt_d2s", dy.data, C, wds[pos:], dx.data, C, dx.data if acc is not None else None, C,
                                 dy.rows, C, C, len(taps), lib.int_array([-shifts[tap] for tap in taps]),
                                 lib.int_array([0] * len(taps)), dy.rows_per_image, dy.hp, dy.wp, plane)
                        pos += len(taps)
                self._put(dy)
            else:
                dz = self._get(nb, Ho, Wo, 4 * C)
                for plane in sorted(planes):
                    taps = planes[plane]
                    ops.gemm_nt(lib.ptr(dy.data), C, wds[pos:], lib.ptr(dz.data[:, plane * C:]), 4 * C, dy.rows, C, C,
                                [-shifts[tap] for tap in taps], [0] * len(taps),
                                rows_per_image=dy.rows_per_image, hp=dy.hp, wp=dy.wp)
                    pos += len(taps)
                self._put(dy)
                lib.call("siss_depth_to_space", dz.data, dx.data, int(acc is not None), nb, x.h, x.w, C)
                self._put(dz)
            self.gmap[id(x)] = dx
        self.tape.append(bwd)
        return y

    def _upsample_subpixel(self, x: Act, pre, cat_with=None):
        """Upsample2D (nearest 2x -> conv3x3) in its SUB-PIXEL form: four 2x2-tap phase convolutions on the low-resolution input
        (optimizer.hip siss_upsample_phase_weights: phase weights = f32 sums of the 3x3 taps, rounded once), 16 instead of 36 tap
        products per low-resolution pixel, no upsampled tensor.  Forward: one product per phase whose epilogue scatters to the
        high-resolution pixels (siss_gemm_nt_d2s_bias).  Backward: space-to-depth of the cotangent, then the dgrad as 16 (plane, tap)
        panels of ONE product and the weight gradient as four 4-panel products into a phase-tap
        scratch that a fold kernel adds onto the nine taps.  Measured per site against upsample copy + persistent 3x3 kernel + fused
        3-tap wgrad (tools/probes/subpixel_upsample.py): 128 -> 256 x 128 ch 1634 -> 1282 us, 64 -> 128 x 256 ch 1309 -> 999 us --
        the flops saved outweigh the generic kernels' lower rate and the consumer GroupNorm's own statistics pass."""
        ps = self.ps
        C, B, lo_h, lo_w = x.c, x.n, x.h, x.w
        H, W = 2 * lo_h, 2 * lo_w
        wname = pre + ".conv.weight"
        if wname not in self._up_w:                     # phase-weight buffers of the sites that take this form (refresh_weights keeps them current)
            _, co_, ci_ = ps.specs[wname].native_shape
            self._up_w[wname] = (torch.empty(4, 4, co_, ci_, dtype=self.adt, device=self.device),
                                 torch.empty(16, ci_, co_, dtype=self.adt, device=self.device))
            lib.call("siss_upsample_phase_weights", ps.p(wname), *self._up_w[wname], co_, ci_)
        wf, wd = self._up_w[wname]
        if cat_with is not None and self.direct_cat:
            assert (cat_with.n, cat_with.h, cat_with.w) == (B, H, W)
            if isinstance(cat_with, ActView):
                assert cat_with.c0 == C and cat_with.base.c == C + cat_with.c
                y = ActView(cat_with.base, 0, C)
            else:
                y = ActView(self._act(self._name("cat"), B, H, W, C + cat_with.c), 0, C)
        else:
            y = self._act(self._name(pre + ".conv"), B, H, W, C)
        y.qstats = None
        y.s2d_cot = self.s2d_from_gn                    # the consuming resnet's norm1 backward may write the cotangent space-to-depth
        wp = x.wp
        ldx = getattr(x, "ld", C)

        def phase_shifts(plane):
            py, px = plane >> 1, plane & 1
            return [(a + py - 1) * wp + (b + px - 1) for a in range(2) for b in range(2)]
        z4 = lib.int_array([0] * 4)
        if self.phase_launch and (not self.f32 or self.f32_fused):    # the four phase products as ONE launch
            lib.call("siss_gemm_nt_d2s_phases", x.data, ldx, wf, y.data, getattr(y, "ld", C), ps.p(pre + ".conv.bias"), None, 0,
                     x.rows, C, C, lib.int_array([0, 4, 8, 12, 16]),
                     lib.int_array([s_ for plane in range(4) for s_ in phase_shifts(plane)]), lib.int_array([0] * 16),
                     x.rows_per_image, x.hp, x.wp)
        else:
            for plane in range(4):
                lib.call("siss_gemm_nt_d2s_bias", x.data, ldx, wf[plane], y.data, getattr(y, "ld", C), ps.p(pre + ".conv.bias"),
                         x.rows, C, C, 4, lib.int_array(phase_shifts(plane)), z4, x.rows_per_image, x.hp, x.wp, plane)

        def bwd():
            nb, gb = self.nb, self.gbase
            dy = self._take(y)
            if (dy.h, dy.w, dy.c) == (lo_h, lo_w, 4 * C):   # already space-to-depth (written so by the consumer's GroupNorm backward)
                z = dy
            else:
                z = self._get(nb, lo_h, lo_w, 4 * C)
                lib.call("siss_space_to_depth_ld", dy.data, z.data, nb, H, W, C, 0)
                self._put(dy)
            # weight gradient: per plane, Y = the plane's columns of z, X = the four shifted low-resolution panels
            rows_per_set = self.set_images * z.rows_per_image
            rb, re = z.wp + 1, rows_per_set - (z.wp + 1)
            zp = ops.zero_page(self.device)
            assert x.n in (nb, self.set_images)
            queued = bool(self.subpixel_queue and self.group_rows and re - rb <= self.group_rows)
            # (queued: the scratch must survive until the grouped launch has run -- one per site, not the shared one)
            dW4 = self._buf((pre if queued else "up") + ".dW4", (self.nsets, 4, 4, C, C))
            dW4.zero_()
            dWt, nsets_ = ps.grads[gb:, ps.specs[wname].off:], self.nsets
            fold = lambda: lib.call("siss_upsample_phase_wgrad_fold", dW4, dWt, ps.total, nsets_, C, C)
            for plane in range(4):                       # (the scratch's set stride is 16 C^2, the bias gradient's the flat buffer's)
                if queued:
                    sh4 = phase_shifts(plane)
                    self._wq.append((lib.TNJob(Y=z.data[:, plane * C:].data_ptr(), ldy=4 * C, X=x.data.data_ptr(), ldx=ldx,
                                               dW=dW4[:, plane].data_ptr(), set_stride=dW4[0].numel(), N=C, C=C, npanels=4,
                                               nsets=self.nsets, rows_per_set=rows_per_set, row_begin=rb, row_end=re, nsplits=0,
                                               x_set_rows=rows_per_set if x.n == nb else 0, zero_page=zp.data_ptr(),
                                               dbias=ps.g(pre + ".conv.bias", gb).data_ptr(), dbias2=None,
                                               shifts=(lib.I * 9)(*sh4, *([0] * 5)), coffs=(lib.I * 9)(*([0] * 9)),
                                               bias_set_stride=ps.total), (z, x)))
                else:
                    lib.call("siss_gemm_tn_bs", z.data[:, plane * C:], 4 * C, x.data, ldx, dW4[:, plane], dW4[0].numel(), C, C, 4,
                             lib.int_array(phase_shifts(plane)), z4, self.nsets, rows_per_set, rows_per_set if x.n == nb else 0,
                             rb, re, 0, zp, ps.g(pre + ".conv.bias", gb), None, ps.total)
            if queued:                                   # the nine-tap fold follows the grouped launch that forms the 16 phase-tap gradients
                self._held[id(z.buf)] = z
                self._wq_post.append(fold)
            else:
                fold()
            # dgrad: dx[Y, X] = sum over (plane, tap) of z_plane[Y - dy_tap, X - dx_tap] . W_plane,tap^T
            dx = self._get(nb, lo_h, lo_w, C)
            sh = [-s_ for plane in range(4) for s_ in phase_shifts(plane)]
            co = [plane * C for plane in range(4) for _ in range(4)]
            ops.gemm_nt(lib.ptr(z.data), 4 * C, wd, lib.ptr(dx.data), C, z.rows, C, C, sh, co,
                        rows_per_image=z.rows_per_image, hp=z.hp, wp=z.wp)
            self._put(z)
            self._give(x, dx)
        self.tape.append(bwd)
        return y

    def upsample(self, x: Act, pre, cat_with=None):
        if self.subpixel_up and (not self.f32 or self.f32_fused) and x.h * x.w >= self.subpixel_min_px:
            return self._upsample_subpixel(x, pre, cat_with)
        ps = self.ps
        C, B = x.c, x.n
        u = self._act(self._name(pre + ".u"), B, 2 * x.h, 2 * x.w, C)
        lib.call("siss_upsample2x", x.data, u.data, B, x.h, x.w, C)
        y, c_b = self.conv(u, pre + ".conv", cat_with=cat_with, want_stats=True)

        def bwd():
            dy = self._take(y)
            du = c_b(dy)
            self._put(dy)
            dx = self._get(self.nb, x.h, x.w, C)
            lib.call("siss_upsample2x_bwd", du.data, dx.data, self.nb, x.h, x.w, C)
            self._put(du)
            self._give(x, dx)
        self.tape.append(bwd)
        return y

    def concat(self, a: Act, b: Act):
        if isinstance(a, ActView) and a.c0 == 0 and a.base.c == a.c + b.c:
            # a's producer wrote it into the head columns already (conv(cat_with=b))
            out = a.base
            assert (b.n, b.h, b.w) == (a.n, a.h, a.w)
            if isinstance(b, ActView):          # ... and the skip was produced in the tail columns: nothing to copy
                assert b.base is out and b.c0 == a.c
            else:
                lib.call("siss_concat_tail", b.data, out.data, a.n, a.h, a.w, a.c, b.c)
        elif isinstance(b, ActView):            # skip in place, head from a producer without a view (attention)
            out = b.base
            assert b.c0 == a.c and out.c == a.c + b.c and not isinstance(a, ActView)
            out.data[:, :a.c].copy_(a.data)
        else:
            out = self._act(self._name("cat"), a.n, a.h, a.w, a.c + b.c)
            lib.call("siss_concat", a.data, b.data, out.data, a.n, a.h, a.w, a.c, b.c)
        out.cat_parts, out.cat_done = (a, b), False

        def bwd():
            nb = self.nb
            if out.cat_done:                    # the consuming resnet's norm1 backward already split the cotangent
                out.cat_done = False
                return
            dcat = self._take(out)
            da = self._get(nb, a.h, a.w, a.c)
            accb = self.gmap.get(id(b))
            self._wsync(accb)
            db = accb if accb is not None else self._get(nb, b.h, b.w, b.c)
            lib.call("siss_concat_bwd", dcat.data, da.data, db.data, int(accb is not None), nb, a.h, a.w, a.c, b.c)
            self.gmap[id(b)] = db
            self._put(dcat)
            self._give(a, da)
        self.tape.append(bwd)
        return out

    # ------------------------------------------------------------------ whole network
    def forward(self, x, t):
        """x: [N, Cin, H, W] f32/bf16 NCHW (device), t: [N] int64.  Returns pred [N, Cout, H, W] f32."""
        with lib.f32_mode(self.f32):
            return self._forward(x.float() if self.f32 else x, t)

    def _forward(self, x, t):
        cfg, ps = self.cfg, self.ps
        assert x.is_cuda and x.dim() == 4 and x.is_contiguous()
        N, cin, H, W = x.shape
        assert cin == cfg.in_channels
        self.tape, self.gmap, self._uid = [], {}, 0
        self._wq, self._held, self._held_release = [], {}, []
        self._pair1 = []
        self._wq_post = []
        self._qs_cache = {}
        if self._wT_stale:                             # the dgrad weight copies of the last optimizer step: beside this forward pass
            st = self._side_stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                self._refresh_dgrad_copies()
            self._prep_pending = True
        self._side_mark = None
        self.nf = N
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        self.time_embed(t)
        heads = self._skip_heads()                     # per skip, in push order: channels of its up-path partner
        h = self._conv_in(x, skip_head=heads[0])
        skips = [h]
        early = self._early_blocks()
        self._early_mark = None
        for (i, cin_b, cout_b, attn, down) in self.plan_down:
            if self._early_mark is None and f"down_blocks.{i}." in early:
                self._early_mark = len(self.tape)      # closures from here on belong to the early-final group
            for j in range(cfg.layers_per_block):
                # a conv-produced skip is written straight into the tail columns of the concat buffer it ends up in
                h = self.resnet(h, f"down_blocks.{i}.resnets.{j}", skip_head=None if attn else heads[len(skips)])
                if attn:
                    h = self.attention(h, f"down_blocks.{i}.attentions.{j}")
                skips.append(h)
            if down:
                h = self.downsample(h, f"down_blocks.{i}.downsamplers.0", skip_head=heads[len(skips)])
                skips.append(h)
        h = self.resnet(h, "mid_block.resnets.0")
        h = self.attention(h, "mid_block.attentions.0")
        h = self.resnet(h, "mid_block.resnets.1", cat_with=skips[-1])
        for (i, cout_b, attn, up, rs) in self.plan_up:
            if self._side_mark is None and h.h * h.w > self.side_max_px:
                self._side_mark = len(self.tape)       # the closures below this index are the low-resolution middle of the backward pass
            for j in range(len(rs)):
                h = self.concat(h, skips.pop())
                # a resnet whose output goes straight into the next concat (no attention / upsample in between)
                # writes it into that concat buffer
                direct = bool(skips) and not attn and not (j == len(rs) - 1 and up)
                h = self.resnet(h, f"up_blocks.{i}.resnets.{j}", cat_with=skips[-1] if direct else None)
                if attn:
                    h = self.attention(h, f"up_blocks.{i}.attentions.{j}")
            if up:
                h = self.upsample(h, f"up_blocks.{i}.upsamplers.0", cat_with=skips[-1] if skips else None)
        assert not skips
        return self._head(h)

    def _skip_heads(self):
        """For every skip connection, in PUSH order: the channel count of the up-path activation it will be concatenated
        behind -- or None when that partner is not written through a view (attention output), in which case the skip
        stays an ordinary activation and concat() copies.  Mirrors the up-path loop of forward()."""
        cfg = self.cfg
        remaining = 1 + sum(cfg.layers_per_block + (1 if down else 0) for (_, _, _, _, down) in self.plan_down)
        heads, hc, direct = [], cfg.block_out_channels[-1], True        # mid_block.resnets.1 writes through a view
        for (i, cout_b, attn, up, rs) in self.plan_up:
            for j in range(len(rs)):
                heads.append(hc if direct and self.direct_cat else None)
                remaining -= 1
                direct = remaining > 0 and not attn and not (j == len(rs) - 1 and up)
                hc = cout_b
            if up:
                direct = remaining > 0                                     # the upsample conv writes through a view
        assert remaining == 0
        return heads[::-1]

    def _conv_in(self, x, skip_head=None):
        """conv_in: im2col rows (K = 9*Cin padded to 64) then a one-panel GEMM."""
        cfg, ps = self.cfg, self.ps
        N, cin, H, W = x.shape
        kp = ps.specs["conv_in.weight"].native_shape[1]
        col = self._act("conv_in.col", N, H, W, kp)
        lib.call("siss_im2col3x3", x, int(x.dtype == torch.bfloat16), col.data, N, cin, H, W, kp, 0)
        c0 = cfg.block_out_channels[0]
        if skip_head is not None and self.direct_cat:
            h = ActView(self._act(self._name("cat"), N, H, W, skip_head + c0), skip_head, c0)
        else:
            h = self._act("conv_in.out", N, H, W, c0)
        ops.gemm_nt(lib.ptr(col.data), kp, ps.sh("conv_in.weight"), lib.ptr(h.data), getattr(h, "ld", c0), col.rows, c0, kp, [0], [0],
                    bias=ps.p("conv_in.bias"), rows_per_image=col.rows_per_image, hp=col.hp, wp=col.wp)
        h0 = h

        def conv_in_bwd():
            dh = self._take(h0)
            dW = ps.grads[self.gbase:, ps.specs["conv_in.weight"].off:]
            self._wgrad(dh, col, dW, c0, kp, 1, dbias=ps.g("conv_in.bias", self.gbase))
            self._put(dh)
        self.tape.append(conv_in_bwd)
        return h

    def _head(self, h: Act):
        """conv_norm_out -> SiLU -> conv_out; returns pred [N, Cout, H, W] f32 and tapes its backward."""
        cfg, ps = self.cfg, self.ps
        N, H, W, c0 = h.n, h.h, h.w, cfg.block_out_channels[0]
        a, gn_b = self.gn(h, "conv_norm_out", True)
        co = cfg.out_channels
        pred = self._buf("pred", (N, co, H, W))
        lib.call("siss_conv_out_fprop", a.data, ps.p("conv_out.weight"), ps.p("conv_out.bias"), pred, N, H, W, c0, co)
        hl = h

        def head_bwd():
            """conv_out backward as GEMMs: the flipped im2col of the (3-channel) cotangent image is a [rows][64]
            matrix `col`; dgrad = col . Wn (one-panel NT GEMM, K = 64), wgrad = col^T . a (one-panel TN GEMM)."""
            c = self.cot
            nb = self.nb
            gb = self.gbase
            kc = self._wd_out.shape[1]
            col = self._get(nb, H, W, kc)
            lib.call("siss_im2col3x3", c, 0, col.data, nb, co, H, W, kc, 1)
            lib.call("siss_nchw_channel_sums", c, self.nsets, self.set_images, co, H * W, ps.total,
                     ps.g("conv_out.bias", gb))
            rows_per_set = self.set_images * col.rows_per_image
            rb, re = col.wp + 1, rows_per_set - (col.wp + 1)
            ns = ops._nsplits(1, 1, self.nsets, re - rb, False)
            if self.pair_top and not self.f32 and re - rb >= self.pair_min_rows:
                zp = ops.zero_page(self.device)
                z9 = (lib.I * 9)(*([0] * 9))
                self._pair1.append((lib.TNJob(Y=col.data.data_ptr(), ldy=kc, X=a.data.data_ptr(), ldx=c0,
                                              dW=ps.grads[gb:, ps.specs["conv_out.weight"].off:].data_ptr(), set_stride=ps.total,
                                              N=9 * co, C=c0, npanels=1, nsets=self.nsets, rows_per_set=rows_per_set, row_begin=rb,
                                              row_end=re, nsplits=0, x_set_rows=rows_per_set if a.n == nb else 0,
                                              zero_page=zp.data_ptr(), dbias=None, dbias2=None, shifts=z9, coffs=z9), col))
                self._held[id(col.buf)] = col
            else:
                lib.call("siss_gemm_tn", col.data, kc, a.data, c0, ps.grads[gb:, ps.specs["conv_out.weight"].off:], ps.total,
                         9 * co, c0, 1, lib.int_array([0]), lib.int_array([0]), self.nsets, rows_per_set,
                         rows_per_set if a.n == nb else 0, rb, re, ns, ops.zero_page(self.device), None, None)
            da = self._get(nb, H, W, c0)
            ops.gemm_nt(lib.ptr(col.data), kc, self._wd_out, lib.ptr(da.data), c0, col.rows, c0, kc, [0], [0],
                        rows_per_image=col.rows_per_image, hp=col.hp, wp=col.wp)
            self._put(col)
            dh = gn_b(da)
            self._put(da)
            self._give(hl, dh)
        self.tape.append(head_bwd)
        return pred

    def zero_grad(self, beside_forward=False, sparse_key=None):
        """beside_forward: the fill runs on the side stream (behind everything issued so far) and is joined by the next backward().
        sparse_key (hashable; the stepper's first micro-batch, with wgrad_overwrite set): everything that decides how the NEXT
        backward pass splits its weight gradients (loss, batch shape).  The first pass under a key runs behind a full fill and
        records which stretches of the gradient buffer it OVERWROTE (siss_gemm_tn_overwrite_log: one-split products under
        nsplits = -2); later fills under the same key skip those stretches (siss_zero_ranges: one launch over the complement),
        and every such pass is checked against the record -- a pass that overwrote anything else raises."""
        self._fill_key, self._fill_plan = None, None
        if not self.fill_grads:
            return
        plan = None
        if sparse_key is not None and self.sparse_fill and self.wgrad_overwrite and not self.f32 and lib.has("siss_zero_ranges"):
            # (+ the schedule switches that decide which launch -- hence which split rule -- a weight gradient takes)
            sparse_key = (sparse_key,) + self._wgrad_sig()
            self._fill_key = sparse_key
            plan = self._fill_plan = self._fill_plans.get(sparse_key)
        g = self.ps.grads

        def fill():
            if plan is None:
                g.zero_()
            elif plan["n"]:
                lib.call("siss_zero_ranges", g, plan["table"], plan["n"], plan["granules"])

        if beside_forward and self.prep_side and self.device.type == "cuda":
            st = self._side_stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                fill()
            self._prep_pending = True
            return
        fill()

    def _wgrad_sig(self):
        return (self.group_rows, self.group_max, self.group_attn, self.pair_top, self.pair_min_rows, self.fold_shortcut, self.subpixel_up,
                self.subpixel_min_px, self.subpixel_queue, self.fused_attn, self.sparse_min_floats, self.wgrad_side, self.side_follow,
                self.on_early_grads_final is not None)

    def _overwritten(self):
        """The drained overwrite log as sorted (first float, floats) stretches of the gradient buffer; None if unusable."""
        recs = lib.overwrite_log()
        if recs is None:
            return None
        g = self.ps.grads
        base, end = g.data_ptr(), g.data_ptr() + g.numel() * 4
        return tuple(sorted(((a - base) // 4, n) for a, n in recs if base <= a and a + 4 * n <= end))

    def _check_sparse_fill(self):
        """End of the backward pass that followed zero_grad(sparse_key=...): record what it overwrote (first pass under the key)
        or compare it with what the sparse fill assumed."""
        key, plan, self._fill_key, self._fill_plan = self._fill_key, self._fill_plan, None, None
        if key is None:
            return
        got = self._overwritten()
        if plan is not None:
            if got != plan["stretches"]:
                self._fill_plans.clear()
                raise RuntimeError("sparse gradient fill: this backward pass did not overwrite the stretches the fill before it skipped "
                                   f"(key {key!r}: {len(plan['stretches'])} recorded, {0 if got is None else len(got)} now) -- the gradients of "
                                   "this step are invalid; schedule switches changed between steps? (UNetEngine.sparse_fill = False)")
            return
        if got is None or torch.cuda.is_current_stream_capturing():    # (the table below is a host-to-device copy)
            return
        # skip list: the overwritten stretches shrunk to 16-byte granules, small ones ignored; the fill covers the complement
        skip, total = [], self.ps.grads.numel()
        for a, n in got:
            a4, b4 = -(-a // 4), (a + n) // 4
            if 4 * (b4 - a4) >= self.sparse_min_floats:
                if skip and a4 < skip[-1][1]:
                    return                                 # overlapping products: not the one-product-per-weight pass this is for
                skip.append((a4, b4))
        starts, lens, at = [], [], 0
        for a4, b4 in skip + [(total // 4, total // 4)]:
            if a4 > at:
                starts.append(at); lens.append(a4 - at)
            at = b4
        assert total % 4 == 0
        pre = [0]
        for n in lens:
            pre.append(pre[-1] + n)
        table = torch.tensor(starts + pre, dtype=torch.int64, device=self.device)
        self._fill_tables.append(table)                    # never freed: a captured step may hold its address (a few KB per key)
        self._fill_plans[key] = dict(stretches=got, n=len(starts), granules=pre[-1], table=table,
                                     skipped_bytes=16 * (total // 4 - pre[-1]))

    def backward(self, cot, nsets=2, grad_base_set=0):
        """cot: [nb, Cout, H, W] f32 cotangent of pred, nb = nsets * set_images.  With the shared
        forward (SISS) nb = 2*N: rows [0,N) seed g_x and rows [N,2N) seed g_a.  Gradients are
        ACCUMULATED into ps.grads[grad_base_set + set] (call zero_grad() at the start of a step)."""
        with lib.f32_mode(self.f32):
            return self._backward(cot, nsets, grad_base_set)

    def _backward(self, cot, nsets, grad_base_set):
        assert cot.is_cuda and cot.dtype == torch.float32 and cot.is_contiguous()
        nb = cot.shape[0]
        assert nb % nsets == 0 and nb % self.nf == 0
        self.nb, self.nsets, self.set_images, self.gbase, self.cot = nb, nsets, nb // nsets, grad_base_set, cot
        if self._wT_stale:                             # (a backward pass without a forward since the last lazy refresh)
            self._refresh_dgrad_copies()
        if self._fill_key is not None:
            lib.overwrite_log(0)                       # (whatever other callers left in the log is not this pass's)
        if self._prep_pending:                         # the gradient fill / dgrad weight copies issued beside the forward pass
            torch.cuda.current_stream().wait_stream(self._side)
            self._prep_pending = False
        d_s = self._buf("temb.d_s", (nb, self.temb_dim))
        d_s.zero_()
        self.dtp_all = self._buf("temb.dtp_all", (nb, self.temb_ntot))
        self.dtp_all.zero_()
        mark = getattr(self, "_early_mark", None)
        side_at = (self._side_mark or 0) - 1 if (self.wgrad_side and not self.f32 and self.nf <= self.side_max_batch) else -1
        for idx in range(len(self.tape) - 1, -1, -1):
            if idx == side_at:
                self._side_phase = True
                self._flush_wgrads_side()               # the weight gradients queued so far run beside the low-resolution blocks
            self.tape[idx]()
            if idx == mark and self.on_early_grads_final is not None:
                self._flush_wgrads()                    # queued low-resolution wgrads belong to the early-final tail
                self._join_side()
                self.on_early_grads_final()         # grads[:, ps.split:] are complete (data-parallel overlap hook)
        self._side_phase = False
        self._flush_wgrads()
        self._join_side()
        self._check_sparse_fill()
        assert not self.gmap, f"{len(self.gmap)} dangling cotangents"
