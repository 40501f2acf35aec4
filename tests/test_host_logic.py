"""CPU tests: C-ABI export table, config loader, samplers, parameter layout maps, DP semantics (gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    h = open(os.path.join(ROOT, "include", "siss_hip.h")).read()
    return sorted(set(re.findall(r"\b(?:int|long) (siss_\w+)\(", h)))


def test_header_is_valid_c():
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "siss_hip.h")])


def test_library_exports_every_declared_symbol():
    from siss_amd import lib
    from siss_amd.build import build
    build(verbose=False)
    so = ctypes.CDLL(lib.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 36
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/siss_hip.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(lib.SIGNATURES) == names
    lib.load()
    # the export count DESIGN.md quotes is the header's (tools/gen_header.py writes it between the markers)
    d = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!--exports-->(\d+) ", d)
    assert m and int(m.group(1)) == len(names), (m and m.group(1), len(names))
    assert lib.abi_version() >= lib.MIN_ABI


def test_ctypes_table_has_the_arity_of_every_prototype():
    """siss_amd/lib.py's argument table against include/siss_hip.h: every entry point is bound with as many arguments as its
    prototype declares (the launchers' trailing `void* stream` included: lib.call appends its VALUE) -- a signature that drifts
    from the header would otherwise only show as garbage arguments on the GPU."""
    from siss_amd import lib
    h = open(os.path.join(ROOT, "include", "siss_hip.h")).read()
    protos = dict(re.findall(r"\b(?:int|long) (siss_\w+)\(([^;{]*?)\)\s*;", h, flags=re.S))
    assert len(protos) == len(lib.SIGNATURES)
    for name, args in protos.items():
        params = [a.strip() for a in args.replace("\n", " ").split(",") if a.strip() and a.strip() != "void"]
        assert len(lib.SIGNATURES[name]) == len(params), (name, len(lib.SIGNATURES[name]), params)


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    from siss_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()


def test_product_package_never_imports_oracle():
    for fn in os.listdir(os.path.join(ROOT, "siss_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "siss_amd", fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


def test_hydra_lite_compose_and_reference_style_features():
    from siss_amd import hydra_lite as H
    c = H.compose("delete_celeb", os.path.join(ROOT, "config"), ["train_batch_size=16", "mixed_precision=bf16"])
    assert c.train_batch_size == 16 and c.mixed_precision == "bf16"
    assert c.dataset_all.data_path == "data/datasets/celeba_hq_256"          # ${data_dir}
    assert list(c.dataset_all.remove_img_names) == ["10000.jpg"]              # ${deletion.img_name}
    assert c.task._target_ == "delete_celeb.DeleteCeleb"
    t = H.compose("delete_tshirt", os.path.join(ROOT, "config"))               # defaults: [train_tshirt_mnist, _self_]
    assert float(t.optimizer.lr) == 5e-5 and list(t.optimizer.betas) == [0.95, 0.999]
    assert t.unet.sample_size == 28 and t.train_batch_size == 64
    o = H.instantiate(c.optimizer)
    assert (o.lr, o.betas, o.weight_decay) == (5e-6, (0.95, 0.999), 1e-6)
    tr = H.instantiate(c.transform)
    img = (np.arange(4 * 4 * 3).reshape(4, 4, 3) * 5).astype(np.uint8)
    x = tr(img)
    assert x.shape == (3, 4, 4) and float(x.min()) >= -1 and float(x.max()) <= 1


def test_samplers_shard_by_rank():
    from siss_amd.data import InfiniteSampler, RepeatedSampler, SyntheticImages, batches
    ds = SyntheticImages(10, (1, 2, 2))
    full = [i for _, i in zip(range(40), InfiniteSampler(ds))]
    r0 = [i for _, i in zip(range(20), InfiniteSampler(ds, rank=0, num_replicas=2))]
    r1 = [i for _, i in zip(range(20), InfiniteSampler(ds, rank=1, num_replicas=2))]
    assert full[0::2] == r0 and full[1::2] == r1
    # golden vectors: the reference's data/utils/infinite_sampler.py run in the build container
    # (len 10 defaults; len 7 rank 1 of 3, seed 4, window 1.0)
    assert full == [2, 8, 4, 9, 1, 6, 2, 3, 0, 9, 7, 8, 1, 5, 6, 4, 3, 8, 2, 1, 9, 6, 7, 5, 8, 4, 0, 9, 6, 7, 3,
                    1, 2, 8, 4, 7, 6, 3, 1, 8]
    ds7 = SyntheticImages(7, (1,))
    got = [i for _, i in zip(range(10), InfiniteSampler(ds7, rank=1, num_replicas=3, seed=4, window_size=1.0))]
    assert got == [6, 1, 2, 2, 4, 4, 2, 5, 4, 3]
    assert list(RepeatedSampler(SyntheticImages(2, (1,)), 3)) == [0, 0, 0, 1, 1, 1]
    b = next(batches(ds, InfiniteSampler(ds), 4))
    assert b.shape == (4, 1, 2, 2)


def test_prefetcher_keeps_the_batch_order_of_the_plain_iterator():
    from siss_amd.data import InfiniteSampler, Prefetcher, SyntheticImages, batches
    ds = SyntheticImages(23, (3, 8, 8), seed=5)
    ref = batches(ds, InfiniteSampler(ds, rank=1, num_replicas=2), 4)
    pf = Prefetcher(ds, InfiniteSampler(ds, rank=1, num_replicas=2), 4, device="cpu", depth=2, workers=3)
    for _ in range(9):
        assert torch.equal(next(pf), next(ref))
    pf.close()

    class Broken(SyntheticImages):
        def __getitem__(self, i):
            raise OSError("unreadable image")
    bad = Prefetcher(Broken(4, (1, 2, 2)), InfiniteSampler(ds), 2, device="cpu")
    try:
        next(bad)
        raise AssertionError("the loader error must surface")
    except RuntimeError as e:
        assert isinstance(e.__cause__, OSError)


def test_param_layout_maps_roundtrip():
    from siss_amd.unet import ParamStore
    ps = ParamStore()
    a = ps.add("c3.weight", "conv3", (8, 16, 3, 3))
    b = ps.add("cin.weight", "conv_in", (8, 3, 3, 3))
    c = ps.add("c1.weight", "conv1", (8, 16, 1, 1))
    g = torch.Generator().manual_seed(0)
    for sp in (a, b, c):
        w = torch.randn(sp.ref_shape, generator=g)
        nat = ParamStore.to_native(sp, w)
        assert tuple(nat.shape) == sp.native_shape
        torch.testing.assert_close(ParamStore.from_native(sp, nat.reshape(-1)), w, rtol=0, atol=0)
    # native conv3 layout is [tap][co][ci]
    w = torch.randn(8, 16, 3, 3, generator=g)
    assert torch.equal(ParamStore.to_native(a, w)[5], w[:, :, 1, 2])


def test_unet_param_registry_matches_checkpoint_shape():
    from siss_amd.config import UNet2DConfig
    from siss_amd.unet import ParamStore, UNetEngine
    e = UNetEngine.__new__(UNetEngine)
    e.cfg, e.ps = UNet2DConfig.celebahq256(), ParamStore()
    e._declare_params()
    assert len(e.ps.specs) == 450
    assert sum(int(np.prod(sp.ref_shape)) for sp in e.ps.specs.values()) == 113_673_219


def test_sd_unet_param_registry_matches_checkpoint_shape():
    """SD v1.5 UNet registry of the HIP engine: 859,520,964 parameters in 686 tensors, same keys and shapes as the
    oracle restatement (which is pinned by the public checkpoint's count, tests/test_oracle_unet.py)."""
    from siss_amd.config import UNet2DConditionConfig
    from siss_amd.unet import ParamStore
    from siss_amd.unet_cond import UNetCondEngine
    from oracle.unet_cond import OracleUNet2DCondition, UNetCondConfig
    e = UNetCondEngine.__new__(UNetCondEngine)
    e.cfg, e.ps = UNet2DConditionConfig.sd15(), ParamStore()
    e._declare_params()
    assert len(e.ps.specs) == 686
    assert sum(int(np.prod(sp.ref_shape)) for sp in e.ps.specs.values()) == 859_520_964
    with torch.device("meta"):
        ref = OracleUNet2DCondition(UNetCondConfig.sd15())
    shapes = {n: tuple(p.shape) for n, p in ref.named_parameters()}
    assert {n: sp.ref_shape for n, sp in e.ps.specs.items()} == shapes


def test_delete_sd_config_composes():
    from siss_amd import hydra_lite as H
    c = H.compose("delete_sd", os.path.join(ROOT, "config"), ["train_batch_size=4", "mixed_precision=bf16"])
    assert c.task._target_ == "delete_sd.DeleteSD" and c.deletion.scaling_norm == 750
    assert c.learning_rate == 1e-5 and c.adam_weight_decay == 1e-2 and c.deletion.loss_params.lambd == 0.5
    assert c.output_dir == "checkpoints/sd/sylvester_stallone"
    import delete_sd
    from siss_amd.tasks import DeleteSD, Task
    assert delete_sd.DeleteSD is DeleteSD and issubclass(DeleteSD, Task)


@pytest.mark.skipif(not os.path.isdir("/root/reference/config"), reason="the reference checkout is only in the build container")
def test_the_references_own_yaml_files_compose_and_instantiate_unchanged():
    """config/*.yaml of this repo are trimmed re-writes; the claim in their header is that the reference's OWN files load unchanged
    through hydra_lite (`--config-path /root/reference/config`): defaults lists, ${} interpolation, the _target_ remap of the
    diffusers / torch / torchvision classes, and the task class resolution (main.py:14-16 of the reference)."""
    from siss_amd import hydra_lite as H
    import delete_celeb, delete_sd, delete_tshirt                           # noqa: E401  (the root shims `task._target_` names)
    ref = "/root/reference/config"
    want = {"delete_celeb": ("delete_celeb.DeleteCeleb", delete_celeb.DeleteCeleb),
            "delete_tshirt": ("delete_tshirt.DeleteTShirt", delete_tshirt.DeleteTShirt),
            "delete_sd": ("delete_sd.DeleteSD", delete_sd.DeleteSD)}
    for name, (target, cls) in want.items():
        c = H.compose(name, ref, ["mixed_precision=bf16"])
        mine = H.compose(name, os.path.join(ROOT, "config"), ["mixed_precision=bf16"])
        assert c.task._target_ == target and H.get_object(target) is cls
        assert c.deletion.loss_fn == mine.deletion.loss_fn and c.deletion.scaling_norm == mine.deletion.scaling_norm
        assert dict(c.deletion.loss_params) == dict(mine.deletion.loss_params)
        assert c.train_batch_size == mine.train_batch_size
        assert c.gradient_accumulation_steps == mine.gradient_accumulation_steps
        if name != "delete_sd":
            o, om = H.instantiate(c.optimizer), H.instantiate(mine.optimizer)
            assert (o.lr, o.betas, o.weight_decay, o.eps) == (om.lr, om.betas, om.weight_decay, om.eps)
            sch = H.instantiate(c.scheduler)
            assert type(sch).__name__ == "DDPMScheduler" and sch.config.num_train_timesteps == 1000
            assert H.instantiate(c.transform) is not None
            assert c.unet.to_dict() == mine.unet.to_dict()
        else:
            for k in ("learning_rate", "adam_beta1", "adam_beta2", "adam_weight_decay", "adam_epsilon", "max_grad_norm", "resolution"):
                assert c[k] == mine[k], k


DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import schedule as S
from oracle.loss import OracleDeletionLoss, siss_terms, mix
from oracle.toy import ToyEps
from siss_amd.dp import (all_gather_params, allreduce_flat_grads, allreduce_pieces, can_shard, recombine_reference,
                         reduce_scatter_param_shards)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
ac = S.alphas_cumprod(); gam, sig = S.gamma_sigma(ac)
g = torch.Generator().manual_seed(123)
Bg, c, hw = 8, 3, 8                                 # GLOBAL batch 8: 4 per rank at N = 2, 1 per rank at N = 8
x0 = torch.rand(Bg, c, hw, hw, generator=g) * 2 - 1
a0 = (torch.rand(1, c, hw, hw, generator=g) * 2 - 1).repeat(Bg, 1, 1, 1)
noise = torch.randn(Bg, c, hw, hw, generator=g); t = torch.full((Bg,), 999); u = torch.rand(Bg, generator=g)
net = ToyEps(c, seed=5)
params = list(net.parameters())
def flat_pair(sl):
    nk, nf = S.add_noise(ac, x0[sl], noise[sl], t[sl]), S.add_noise(ac, a0[sl], noise[sl], t[sl])
    xm = mix(nk, nf, u[sl] > 0.5)
    pred = net(xm, t[sl])[0]
    ex, ea, _, _, iwx, iwa = siss_terms(xm, x0[sl], a0[sl], gam[t[sl]], sig[t[sl]], 0.5)
    lx = (iwx[:, None, None, None] * (pred - ex) ** 2).sum() / Bg      # normaliser = GLOBAL batch
    la = (iwa[:, None, None, None] * (pred - ea) ** 2).sum() / Bg
    gx = torch.autograd.grad(lx, params, retain_graph=True); ga = torch.autograd.grad(la, params)
    fp = torch.stack([torch.cat([v.flatten() for v in gx]), torch.cat([v.flatten() for v in ga])])
    pad = (-fp.shape[1]) % (4 * 8)                    # the flat buffers are 64-float aligned in the engine: shards at N = 8 too
    return torch.cat([fp, torch.zeros(2, pad)], 1).contiguous()
per = Bg // world
local = flat_pair(slice(rank * per, (rank + 1) * per))
P = local.shape[1]
whole = flat_pair(slice(0, Bg))                       # single-process global-batch oracle

# (1) the serial exchange: one all-reduce of the flat pair = the global-batch gradients, replicas identical
mine = local.clone()
allreduce_flat_grads(mine)
torch.testing.assert_close(mine, whole, rtol=1e-4, atol=1e-5)      # f32 sums in a different order
g1, s1 = recombine_reference(mine[0], mine[1], 5.0); g2, s2 = recombine_reference(whole[0], whole[1], 5.0)
torch.testing.assert_close(g1, g2, rtol=1e-4, atol=1e-5)
gathered = [torch.zeros_like(g1) for _ in range(world)]; dist.all_gather(gathered, g1)
assert all(torch.equal(gathered[0], q) for q in gathered)   # replicas stay identical

# (2) the overlapped exchange: TWO grouped collectives -- [tail_x, tail_a] (async, from inside the backward), then
#     [head_x, head_a] -- give the same sums as the one all-reduce, for any split point
for split in (0, 7, P // 3, P):
    ov = local.clone()
    pend = [allreduce_pieces([ov[k, split:] for k in range(2)], None, async_op=True)] if split < P else []
    if split > 0:
        pend.append(allreduce_pieces([ov[k, :split] for k in range(2)], None, async_op=True))
    for w in pend:
        w.wait()
    torch.testing.assert_close(ov, mine, rtol=1e-5, atol=1e-5)           # (gloo sums pieces in another ring order)
    both = [torch.zeros_like(ov) for _ in range(world)]; dist.all_gather(both, ov)
    assert all(torch.equal(both[0], q) for q in both), split

# (3) the sharded update: reduce-scatter -> shard-local norm sums (3 doubles all-reduced) / recombine / clip / AdamW ->
#     all-gather of the parameters.  Slice arithmetic at this world size; same update as the replicated one; moments live on
#     the owner's shard until they are gathered (mode switch / checkpoint).
assert can_shard(P, world) and not can_shard(P + 2, world) and not can_shard(P, 1)
def adamw(p, m, v, gx, ga, sums, step, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2, sn=5.0, max_norm=1.0):
    nx2, na2, dot = (float(q) for q in sums)
    s = sn / na2 ** 0.5
    pre = max(nx2 - 2 * s * dot + s * s * na2, 0.0) ** 0.5
    coef = min(1.0, max_norm / (pre + 1e-6))
    gg = (gx - s * ga) * coef
    m.mul_(b1).add_(gg, alpha=1 - b1); v.mul_(b2).addcmul_(gg, gg, value=1 - b2)
    p.mul_(1 - lr * wd).addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)
p_rep = torch.linspace(-1, 1, P).double(); m_rep, v_rep = torch.zeros(P).double(), torch.zeros(P).double()
p_sh, m_sh, v_sh = p_rep.clone(), m_rep.clone(), v_rep.clone()
p0 = p_rep.clone()
full = mine.double()
for step in (1, 2):
    sums = torch.stack([full[0] @ full[0], full[1] @ full[1], full[0] @ full[1]])
    adamw(p_rep, m_rep, v_rep, full[0], full[1], sums, step)                 # replicated update on the all-reduced pair
    gx_s, ga_s, lo, hi = reduce_scatter_param_shards(local.clone())
    assert (lo, hi) == (rank * P // world, (rank + 1) * P // world)
    torch.testing.assert_close(gx_s, mine[0, lo:hi], rtol=1e-4, atol=1e-5)     # rank-order sum vs gloo's ring order
    part = torch.stack([gx_s.double() @ gx_s.double(), ga_s.double() @ ga_s.double(), gx_s.double() @ ga_s.double()])
    dist.all_reduce(part)
    adamw(p_sh[lo:hi], m_sh[lo:hi], v_sh[lo:hi], gx_s.double(), ga_s.double(), part, step)
    all_gather_params(p_sh, lo, hi)
    du_s, du_r = p_sh - p0, p_rep - p0                 # same update (AdamW's first steps are sign-like: compare directions)
    assert float((du_s @ du_r) / (du_s.norm() * du_r.norm())) > 0.9999 and float((du_s - du_r).abs().max()) < 2e-3
    both = [torch.zeros_like(p_sh) for _ in range(world)]; dist.all_gather(both, p_sh)
    assert all(torch.equal(both[0], q) for q in both), "sharded update: replicas diverged"
# outside its shard a rank's moments are stale (still zero) ...
other = (rank + 1) % world
assert float(m_sh[other * P // world:(other + 1) * P // world].abs().max()) == 0.0
# ... until they are gathered: then every rank holds the replicated moments
all_gather_params(m_sh, lo, hi); all_gather_params(v_sh, lo, hi)
torch.testing.assert_close(m_sh, m_rep, rtol=1e-3, atol=1e-7); torch.testing.assert_close(v_sh, v_rep, rtol=1e-3, atol=1e-10)
for buf in (m_sh, v_sh):
    both = [torch.zeros_like(buf) for _ in range(world)]; dist.all_gather(both, buf)
    assert all(torch.equal(both[0], q) for q in both), "gathered moments differ between ranks"
dist.destroy_process_group()
print("dp ok", rank)
'''


@pytest.mark.parametrize("world", [2, 8])
def test_dp_ranks_equal_global_batch(tmp_path, world):
    """N = 2 and N = 8 gloo ranks on the CPU: serial all-reduce, the overlapped exchange's two grouped collectives and the
    sharded update (slice arithmetic, moment re-gather) against the single-process global batch."""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    port = str(29533 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("dp ok") == world


# ---------------------------------------------------------------------------------------------------------------------
# f-3: the data-input samplers, pinned by index sequences recorded from the REFERENCE's own classes
# (tests/golden/samplers.npz, made by oracle/make_golden.py::sampler_cases from data/utils/infinite_sampler.py:4-35 and
# data/utils/repeat_sampler.py:4-21).  Bit-exact (integer work).
# ---------------------------------------------------------------------------------------------------------------------
def test_samplers_reproduce_the_reference_index_sequences(golden_dir):
    import itertools
    import re as _re
    from siss_amd.data import InfiniteSampler, RepeatedSampler
    z = np.load(os.path.join(golden_dir, "samplers.npz"))
    seen = 0
    for key in z.files:
        want = z[key]
        m = _re.fullmatch(r"inf_n(\d+)_r(\d+)of(\d+)_sh(\d)_s(\d+)_w([\d.]+)", key)
        if m:
            n, rank, world, sh, seed = (int(m.group(i)) for i in range(1, 6))
            smp = InfiniteSampler(list(range(n)), rank=rank, num_replicas=world, shuffle=bool(sh), seed=seed,
                                  window_size=float(m.group(6)))
            got = np.fromiter(itertools.islice(iter(smp), want.size), dtype=np.int64, count=want.size)
        else:
            m = _re.fullmatch(r"rep_n(\d+)_x(\d+)", key)
            assert m, key
            smp = RepeatedSampler(list(range(int(m.group(1)))), int(m.group(2)))
            got = np.array(list(iter(smp)), dtype=np.int64)
            assert len(smp) == want.size
        assert got.shape == want.shape and np.array_equal(got, want), key
        seen += 1
    assert seen >= 12
    # the shards of one global sequence: ranks r of R interleave back to the single-process order
    one = z["inf_n1000_r0of1_sh1_s0_w0.5"]
    assert np.array_equal(z["inf_n1000_r0of8_sh1_s0_w0.5"][:1250], one[0::8])
    assert np.array_equal(z["inf_n1000_r3of8_sh1_s0_w0.5"][:1250], one[3::8])


# ---------------------------------------------------------------------------------------------------------------------
# f-2: unet/config.json as diffusers 0.27.2 writes it (every key, defaults included) loads; a config that describes a
# network the HIP engine does not implement is REFUSED instead of silently computing the default architecture.
# ---------------------------------------------------------------------------------------------------------------------
DIFFUSERS_UNET2D_JSON = {
    "_class_name": "UNet2DModel", "_diffusers_version": "0.27.2", "_name_or_path": "google/ddpm-celebahq-256",
    "act_fn": "silu", "add_attention": True, "attention_head_dim": None, "attn_norm_num_groups": None,
    "block_out_channels": [128, 128, 256, 256, 512, 512], "center_input_sample": False, "class_embed_type": None,
    "down_block_types": ["DownBlock2D"] * 4 + ["AttnDownBlock2D", "DownBlock2D"], "downsample_padding": 0,
    "downsample_type": "conv", "dropout": 0.0, "flip_sin_to_cos": False, "freq_shift": 1, "in_channels": 3,
    "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-06, "norm_num_groups": 32, "num_class_embeds": None,
    "num_train_timesteps": None, "out_channels": 3, "resnet_time_scale_shift": "default", "sample_size": 256,
    "time_embedding_type": "positional", "up_block_types": ["UpBlock2D", "AttnUpBlock2D"] + ["UpBlock2D"] * 4,
    "upsample_type": "conv",
}


def test_diffusers_config_json_loads_and_unsupported_architectures_are_refused():
    from siss_amd.config import UNet2DConditionConfig, UNet2DConfig
    c = UNet2DConfig.from_dict(DIFFUSERS_UNET2D_JSON)
    assert c == UNet2DConfig.celebahq256()
    for key, val in (("resnet_time_scale_shift", "scale_shift"), ("add_attention", False), ("time_embedding_type", "fourier"),
                     ("center_input_sample", True), ("mid_block_scale_factor", 2.0), ("dropout", 0.1),
                     ("downsample_type", "resnet"), ("class_embed_type", "timestep"), ("act_fn", "gelu"),
                     ("attn_norm_num_groups", 8), ("some_future_key", 1)):
        with pytest.raises(ValueError, match=key):
            UNet2DConfig.from_dict({**DIFFUSERS_UNET2D_JSON, key: val})
    with pytest.raises(ValueError, match="down_block_types"):
        UNet2DConfig.from_dict({**DIFFUSERS_UNET2D_JSON, "down_block_types": ["SkipDownBlock2D"] * 6})
    sd = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=[320, 640, 1280, 1280],
              down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3,
              layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
              downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, act_fn="silu", center_input_sample=False,
              mid_block_type="UNetMidBlock2DCrossAttn", only_cross_attention=False, use_linear_projection=False,
              upcast_attention=False, dual_cross_attention=False, class_embed_type=None, num_class_embeds=None,
              resnet_time_scale_shift="default", mid_block_scale_factor=1, _class_name="UNet2DConditionModel")
    assert UNet2DConditionConfig.from_dict(sd) == UNet2DConditionConfig.sd15()
    for key, val in (("use_linear_projection", True), ("only_cross_attention", True), ("transformer_layers_per_block", 2),
                     ("addition_embed_type", "text_time")):       # SD 2.x / SDXL UNets are different networks
        with pytest.raises(ValueError, match=key):
            UNet2DConditionConfig.from_dict({**sd, key: val})


def test_lr_multiplier_matches_the_published_schedules():
    """diffusers.optimization.get_scheduler as delete_celeb.py:296-301 builds it (cfg.lr_scheduler, warmup_steps,
    training_steps), checked against torch's LambdaLR driving the published lambdas."""
    import math
    from siss_amd.scheduler import lr_multiplier
    W, T = 3, 10
    lam = {"constant": lambda s: 1.0, "constant_with_warmup": lambda s: s / max(1.0, W) if s < W else 1.0,
           "linear": lambda s: s / max(1, W) if s < W else max(0.0, (T - s) / max(1, T - W)),
           "cosine": lambda s: s / max(1, W) if s < W else max(0.0, 0.5 * (1 + math.cos(math.pi * (s - W) / max(1, T - W))))}
    for name, f in lam.items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1.0)
        sch = torch.optim.lr_scheduler.LambdaLR(opt, f)
        for s in range(T + 2):
            assert abs(opt.param_groups[0]["lr"] - lr_multiplier(name, s, W, T)) < 1e-12, (name, s)
            opt.step(); sch.step()
    with pytest.raises(NotImplementedError):
        lr_multiplier("polynomial", 0, W, T)


def test_delete_sd_schedule_args_noise_offset_and_refusals(monkeypatch):
    """delete_sd.py:714-719 builds its schedule from lr_warmup_steps * num_processes (config/delete_sd.yaml has no
    `warmup_steps` key); :546-552 scales the LR; :893-898 adds offset noise; the knobs the loop does not implement
    (input_perturbation, snr_gamma, a non-epsilon prediction_type, 8-bit Adam) are refused, not ignored."""
    from siss_amd import hydra_lite as H
    from siss_amd.tasks import DeleteCeleb, DeleteSD
    cfgdir = os.path.join(ROOT, "config")
    c = H.compose("delete_sd", cfgdir, ["lr_warmup_steps=5", "lr_scheduler=constant_with_warmup"])
    t = DeleteSD(c)
    assert t.lr_schedule_args(world=1) == (5, int(c.training_steps))
    assert t.lr_schedule_args(world=8) == (40, int(c.training_steps))
    cc = H.compose("delete_celeb", cfgdir, ["warmup_steps=7"])
    assert DeleteCeleb(cc).lr_schedule_args(world=8) == (7, int(cc.training_steps))      # delete_celeb.py:296-301: no world factor
    t.check_supported()                                                                  # the shipped values pass
    # scale_lr (delete_sd.py:546-552)
    lr0 = t.optimizer_args()[0]
    monkeypatch.setenv("WORLD_SIZE", "4")
    cs = H.compose("delete_sd", cfgdir, ["scale_lr=true", "train_batch_size=2", "gradient_accumulation_steps=3"])
    assert abs(DeleteSD(cs).optimizer_args()[0] - lr0 * 3 * 2 * 4) < 1e-18
    monkeypatch.delenv("WORLD_SIZE")
    # offset noise: plain noise + offset * one draw per (sample, channel), same generator stream
    co = H.compose("delete_sd", cfgdir, ["noise_offset=0.1"])
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    got = DeleteSD(co).sample_noise((2, 4, 8, 8), "cpu", g1)
    base = torch.randn((2, 4, 8, 8), generator=g2)
    want = base + 0.1 * torch.randn((2, 4, 1, 1), generator=g2)
    assert torch.equal(got, want)
    assert torch.equal(t.sample_noise((2, 4, 8, 8), "cpu", torch.Generator().manual_seed(3)), base)
    for ov, pat in (("input_perturbation=0.1", "input_perturbation"), ("snr_gamma=5.0", "snr_gamma"),
                    ("prediction_type=v_prediction", "prediction_type"), ("use_8bit_adam=true", "use_8bit_adam")):
        with pytest.raises(NotImplementedError, match=pat):
            DeleteSD(H.compose("delete_sd", cfgdir, [ov])).check_supported()


def test_forced_collectives_on_a_one_rank_group_are_identities(monkeypatch):
    """The test-only switch dp.FORCE_COLLECTIVES (used by tests/test_hip_rccl.py for RCCL's world-size-1 first contact): with one
    rank the collectives are normally skipped; forced, they run on the backend and must return their input (gloo here)."""
    import torch.distributed as dist
    from siss_amd import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        t = torch.arange(64, dtype=torch.float32).view(2, 32).clone()
        want = t.clone()
        calls = []
        orig = dist.all_reduce
        monkeypatch.setattr(dist, "all_reduce", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        monkeypatch.setattr(dp, "FORCE_COLLECTIVES", False)
        assert not dp.active() and dp.allreduce_pieces([t[0], t[1]]) is None and not dp.can_shard(32, 1)
        dp.allreduce_flat_grads(t)
        assert not calls
        monkeypatch.setattr(dp, "FORCE_COLLECTIVES", True)
        assert dp.active() and dp.can_shard(32, 1)
        dp.allreduce_flat_grads(t)
        w = dp.allreduce_pieces([t[0, 16:], t[1, 16:]], async_op=True)
        w.wait()
        assert len(calls) == 3 and torch.equal(t, want)
        gx, ga, lo, hi = dp.reduce_scatter_param_shards(t)
        assert (lo, hi) == (0, 32) and torch.equal(gx, want[0]) and torch.equal(ga, want[1])
        flat = want[0].clone()
        dp.all_gather_params(flat, lo, hi)
        assert torch.equal(flat, want[0])
    finally:
        dist.destroy_process_group()
