"""Worker for tests/test_hip_dp.py: 2 ranks (gloo, both on cuda:0 -- RCCL refuses two ranks on one device)
run one SISS step each on its own shard through SISSStepper(process_group=WORLD); rank 0 also runs the
single-process step on the concatenated global batch and compares the updated parameters."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, sys.argv[1])
from siss_amd.config import UNet2DConfig      # noqa: E402
from siss_amd.step import SISSStepper          # noqa: E402
from siss_amd.unet import UNetEngine           # noqa: E402

KW = dict(sample_size=16, in_channels=3, out_channels=3, block_out_channels=(64, 128),
          down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"),
          layers_per_block=1, attention_head_dim=None, norm_num_groups=32, norm_eps=1e-6,
          downsample_padding=0, flip_sin_to_cos=False, freq_shift=1)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000), 0)
    g = torch.Generator().manual_seed(99)
    Bg = 4
    x0 = torch.rand(Bg, 3, 16, 16, generator=g) * 2 - 1
    a0 = (torch.rand(1, 3, 16, 16, generator=g) * 2 - 1).repeat(Bg, 1, 1, 1)
    noise = torch.randn(Bg, 3, 16, 16, generator=g)
    t = torch.full((Bg,), 999, dtype=torch.long)
    u = torch.rand(Bg, generator=g)
    kw = dict(lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, scaling_norm=5.0, lambd=0.5, mixed_precision=None)

    eng = UNetEngine(UNet2DConfig(**KW), dev)
    sd = eng.init_random(seed=7)
    per = Bg // world
    sl = slice(rank * per, (rank + 1) * per)
    st = SISSStepper(eng, ac, train_batch_size=per, process_group=dist.group.WORLD, **kw)
    st.step(x0[sl], a0[sl], noise[sl], t[sl].to(dev), u[sl])
    dp_stats = st.stats()
    dp_params = eng.ps.flat.clone()

    # replicas must hold identical parameters after the step
    other = [torch.zeros_like(dp_params) for _ in range(world)]
    dist.all_gather(other, dp_params)
    assert all(torch.equal(other[0], o) for o in other), "replicas diverged"

    if rank == 0:
        eng.load_state_dict(sd)
        ref = SISSStepper(eng, ac, train_batch_size=Bg, process_group=None, **kw)
        ref.step(x0, a0, noise, t.to(dev), u)
        rs = ref.stats()
        for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm"):
            assert abs(dp_stats[k] - rs[k]) <= 2e-2 * abs(rs[k]), (k, dp_stats[k], rs[k])
        upd_ref = eng.ps.flat.clone()
        # update-direction cosine >= 0.99 over the elements with a significant gradient
        # (|g| > 1e-6 |g|_inf; AdamW's first step is sign-like, numerically-zero gradients move at random)
        g_fin = eng.ps.grads[0] - rs["scaling_factor"] * eng.ps.grads[1]
        mask = g_fin.abs() > 1e-6 * g_fin.abs().max()
        eng.load_state_dict(sd)
        base = eng.ps.flat.clone()
        du_ref, du_dp = (upd_ref - base)[mask].double(), (dp_params - base)[mask].double()
        cos = float((du_ref * du_dp).sum() / (du_ref.norm() * du_dp.norm()))
        assert float(mask.float().mean()) > 0.5 and cos >= 0.99, (cos, float(mask.float().mean()))
    # serial exchange (one all-reduce after the backward) gives the same update as the overlapped one ...
    eng.load_state_dict(sd)
    st2 = SISSStepper(eng, ac, train_batch_size=per, process_group=dist.group.WORLD, **kw)
    assert st2.overlap, "the overlapped exchange is the default when the flat layout has an early-final tail"
    st2.set_overlap(False)
    st2.step(x0[sl], a0[sl], noise[sl], t[sl].to(dev), u[sl])
    du_serial = eng.ps.flat.clone()
    base = torch.empty_like(du_serial)
    eng.load_state_dict(sd)
    base.copy_(eng.ps.flat)
    d1, d2 = dp_params - base, du_serial - base
    cos = float((d1 * d2).sum() / (d1.norm() * d2.norm()))
    assert cos > 0.999, cos
    # ... and the measured choice between the two runs on every rank without deadlock and agrees across ranks
    st3 = SISSStepper(eng, ac, train_batch_size=per, process_group=dist.group.WORLD, **kw)
    before = eng.ps.flat.clone()
    choice = st3.autotune_overlap(lambda: st3.step(x0[sl], a0[sl], noise[sl], t[sl].to(dev), u[sl]), iters=1)
    # the timed steps were real optimizer steps: parameters, moments and the step counter are put back afterwards
    assert torch.equal(eng.ps.flat, before) and float(st3.opt.m.abs().max()) == 0.0 and int(st3.opt.scalars[6]) == 0
    flags = [None] * world
    dist.all_gather_object(flags, bool(choice))
    keys = set(st3.overlap_timings)
    # three candidates a node can tell apart: serial all-reduce, overlapped all-reduce, sharded update (when P shards evenly)
    from siss_amd.dp import can_shard
    want = {"overlap_ms", "serial_ms"} | ({"serial_sharded_ms"} if can_shard(eng.ps.total, world) else set())
    assert len(set(flags)) == 1 and keys == want, (keys, want)
    # the stepper's own state is put back too (NegGrad's decaying superfactor, the last statistics)
    eng.load_state_dict(sd)
    kw_ng = dict(kw, loss_fn="simple_neg_del")
    st_ng = SISSStepper(eng, ac, train_batch_size=per, process_group=dist.group.WORLD, superfactor=2.0, superfactor_decay=0.5, **kw_ng)
    st_ng.autotune_overlap(lambda: st_ng.step(x0[sl], a0[sl], noise[sl], t[sl].to(dev), u[sl]), iters=1)
    assert st_ng.superfactor == 2.0 and st_ng.last is None and st_ng._micro == 0
    if "serial_sharded_ms" in keys:
        # the SHARDED update (reduce-scatter -> this rank's half of norm-fix / clip / AdamW -> all-gather of the parameters,
        # SURVEY.md section 5) is the same step: same scalars, same update, replicas bit-identical
        eng.load_state_dict(sd)
        st5 = SISSStepper(eng, ac, train_batch_size=per, process_group=dist.group.WORLD, **kw)
        st5.set_overlap(False, "sharded")
        st5.step(x0[sl], a0[sl], noise[sl], t[sl].to(dev), u[sl])
        s5 = st5.stats()
        for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm"):
            assert abs(s5[k] - dp_stats[k]) <= 1e-3 * abs(dp_stats[k]), (k, s5[k], dp_stats[k])
        p5 = eng.ps.flat.clone()
        both = [torch.zeros_like(p5) for _ in range(world)]
        dist.all_gather(both, p5)
        assert all(torch.equal(both[0], o) for o in both), "sharded update: replicas diverged"
        assert torch.equal(eng.ps.shadow, eng.ps.flat.to(torch.bfloat16)), "bf16 operand shadow must follow the gathered master"
        d5 = p5 - base
        cos = float((d5 * d2).sum() / (d5.norm() * d2.norm()))
        assert cos > 0.999, cos
        # Leaving the sharded mode WITHOUT naming a new exchange (set_overlap(True): ADVICE r02) -- the replicated update that
        # follows gathers AdamW's moments itself (each rank advanced only its shard) and keeps the replicas bit-identical
        st5.set_overlap(True)
        st5.step(x0[sl], a0[sl], noise[sl], t[sl].to(dev), u[sl])
        for buf in (eng.ps.flat, st5.opt.m, st5.opt.v):
            both = [torch.zeros_like(buf) for _ in range(world)]
            dist.all_gather(both, buf.clone())
            assert all(torch.equal(both[0], o) for o in both), "replicas diverged after switching from the sharded update"
        # ... and so does a checkpoint of the optimizer taken right after a sharded step
        st5.set_overlap(False, "sharded")
        st5.step(x0[sl], a0[sl], noise[sl], t[sl].to(dev), u[sl])
        m_all, v_all, _ = st5.optimizer_state()
        for buf in (m_all, v_all):
            both = [torch.zeros_like(buf) for _ in range(world)]
            dist.all_gather(both, buf.clone())
            assert all(torch.equal(both[0], o) for o in both), "optimizer_state() must return complete moments on every rank"
    print("dp exchange timings", rank, st3.overlap_timings)
    dist.barrier()
    dist.destroy_process_group()
    print("dp gpu ok", rank)


if __name__ == "__main__":
    main()
