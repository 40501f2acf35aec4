"""Worker for tests/test_hip_rccl.py: RCCL's first contact, on the ONE GPU a build box has.

A world-size-1 `nccl` (= RCCL on ROCm) process group is created in THIS fresh process exactly as bench.py creates it for N > 1
(`init_process_group("nccl", device_id=...)`, bench.py main()), and dp.FORCE_COLLECTIVES makes the step issue its collectives
although one rank has nothing to exchange.  A sum over one rank is the identity, so every result is checked against the step
without a group.  What this proves: librccl loads, the communicator initialises with `device_id`, the collectives are ordered
correctly against the engine's kernels (which run on torch's current stream; RCCL runs on its own), the coalescing window of
dp.allreduce_pieces and the async overlap hook inside UNetEngine.backward work on the real backend -- everything except link
bandwidth.  Replaces /root/reference/delete_celeb.py:99-101,304 (accelerate -> DDP -> NCCL).

    python tests/rccl_ws1_worker.py <repo root> steps     # (i) flat-pair all-reduce, (ii) overlap hook, (iii) steps in all exchange modes
    python tests/rccl_ws1_worker.py <repo root> graph     # (iv) ATTEMPT to capture a step with its collective into a hipGraph

Prints one line `RCCL_WS1 {json}`; also written to gpurun_out/rccl_ws1_<mode>.json when that directory exists.
"""
import json
import os
import socket
import sys
import time

ROOT = sys.argv[1]
MODE = sys.argv[2]
sys.path.insert(0, ROOT)
os.environ["SISS_DP_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402

from siss_amd import dp                        # noqa: E402
from siss_amd.config import UNet2DConfig       # noqa: E402
from siss_amd.step import SISSStepper          # noqa: E402
from siss_amd.unet import UNetEngine           # noqa: E402

KEYS = ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm")
OUT = {"mode": MODE}
CALLS = {"all_reduce": 0, "all_to_all_single": 0, "all_gather_into_tensor": 0}


def count(name):
    orig = getattr(dist, name)

    def wrapped(*a, **k):
        CALLS[name] += 1
        return orig(*a, **k)
    setattr(dist, name, wrapped)


def finish(ok=True):
    OUT["collective_calls"] = dict(CALLS)
    OUT["ok"] = ok
    line = "RCCL_WS1 " + json.dumps(OUT)
    print(line, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, f"rccl_ws1_{MODE}.json"), "w") as f:
            f.write(json.dumps(OUT, indent=1) + "\n")


def batch(B, hw, g, dev):
    x0 = (torch.rand(B, 3, hw, hw, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
    a0 = (torch.rand(1, 3, hw, hw, generator=g) * 2 - 1).repeat(B, 1, 1, 1).to(dev).to(torch.bfloat16)
    noise = torch.randn(B, 3, hw, hw, generator=g).to(dev).to(torch.bfloat16)
    t = torch.full((B,), 999, dtype=torch.long, device=dev)
    u = torch.rand(B, generator=g).to(dev)
    return x0, a0, noise, t, u


def same_step(a, b, what, tol=1e-4):
    """Two runs of one step differ by the order of the weight gradients' f32 atomics only."""
    for k in KEYS:
        assert abs(a[k] - b[k]) <= tol * abs(b[k]), (what, k, a[k], b[k])


def main():
    assert torch.cuda.is_available()
    assert dp.FORCE_COLLECTIVES
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", device_id=dev)               # exactly bench.py's call for N > 1
    pg = dist.group.WORLD
    OUT["backend"] = dist.get_backend()
    OUT["world_size"] = dist.get_world_size()
    OUT["init_s"] = round(time.perf_counter() - t0, 2)
    try:
        OUT["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                                       # informational only
        OUT["rccl_version"] = f"n/a ({type(e).__name__})"
    assert OUT["backend"] == "nccl" and OUT["world_size"] == 1
    for name in CALLS:
        count(name)
    kw = dict(lr=5e-6, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6, scaling_norm=500.0, lambd=0.5, mixed_precision="bf16")
    ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32), 0)
    g = torch.Generator().manual_seed(5)

    if MODE == "steps":
        # the REAL network of BASELINE configs[1]/[2]: 113.7 M parameters, flat pair = 2 x 454.7 MB f32
        B = 2
        eng = UNetEngine(UNet2DConfig.celebahq256(), dev)
        eng.init_random(seed=42)
        p0 = eng.ps.flat.clone()
        mb = batch(B, 256, g, dev)

        def restore(st):
            eng.ps.flat.copy_(p0)
            eng.refresh_weights(cast_shadow=True)
            st.opt.m.zero_(); st.opt.v.zero_(); st.opt.scalars.zero_()

        # reference: the step WITHOUT a group
        ref = SISSStepper(eng, ac, train_batch_size=B, process_group=None, **kw)
        ref.step(*mb)
        ref_stats = ref.stats()
        ref_params = eng.ps.flat.clone()
        assert sum(CALLS.values()) == 0

        # (i) ONE all-reduce of the real flat [g_x ; g_a] pair, in place, on the gradients that step left
        grads = eng.ps.grads
        want = grads.clone()
        dp.allreduce_flat_grads(grads, pg)
        torch.cuda.synchronize()
        assert CALLS["all_reduce"] == 1
        assert torch.equal(grads, want), "a sum over one rank must return the buffer bit for bit"
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for i in range(5):
            ev[i].record()
            dp.allreduce_flat_grads(grads, pg)
        ev[5].record()
        torch.cuda.synchronize()
        assert torch.equal(grads, want)
        OUT["flat_pair"] = {"bytes": grads.numel() * 4, "params": grads.shape[1],
                            "allreduce_ms_1rank": round(min(ev[i].elapsed_time(ev[i + 1]) for i in range(5)), 3)}
        del want

        # (ii) + (iii): the stepper with the group -- default mode = OVERLAP: the tails' grouped collective is issued async from the
        # hook inside UNetEngine.backward, the heads' after it; then the serial all-reduce; then the sharded update
        st = SISSStepper(eng, ac, train_batch_size=B, process_group=pg, **kw)
        assert st.dp_on and st.overlap and eng.on_early_grads_final is not None
        modes = {}
        for name, (ov, exch) in {"overlap": (True, "allreduce"), "serial": (False, "allreduce"),
                                 "serial_sharded": (False, "sharded")}.items():
            restore(st)
            st.set_overlap(ov, exch)
            before = dict(CALLS)
            hooked = []
            if ov:                                   # the hook must fire from INSIDE the backward (before the last tape entries ran)
                inner = eng.on_early_grads_final

                def spy():
                    hooked.append(len(st._pending))
                    inner()
                    hooked.append(len(st._pending))
                eng.on_early_grads_final = spy
            st.step(*mb)
            got = st.stats()
            torch.cuda.synchronize()
            if ov:
                assert hooked == [0, 1], hooked      # one async grouped collective was started by the hook
                eng.on_early_grads_final = inner
            same_step(got, ref_stats, name)
            d = (eng.ps.flat - ref_params).abs()
            frac_same = float((d == 0).float().mean())
            assert float(d.max()) <= 2.5 * kw["lr"] and frac_same > 0.98, (name, float(d.max()), frac_same)
            if name == "serial_sharded":             # moments live on the shard [0, P): gather is a no-op copy on one rank
                st.optimizer_state()
            modes[name] = {"calls": {k: CALLS[k] - before[k] for k in CALLS}, "params_bit_equal_frac": round(frac_same, 5),
                           **{k: got[k] for k in KEYS}}
        assert modes["overlap"]["calls"]["all_reduce"] == 4          # 2 pieces x 2 grouped collectives
        assert modes["serial"]["calls"]["all_reduce"] == 1
        assert modes["serial_sharded"]["calls"]["all_to_all_single"] == 2 and modes["serial_sharded"]["calls"]["all_gather_into_tensor"] >= 1
        OUT["modes"] = modes
        OUT["no_group"] = {k: ref_stats[k] for k in KEYS}

        # the autotune (all_gather_object, barrier, MAX all-reduce, state restore) on the real backend
        restore(st)
        st.set_overlap(True, "allreduce")
        st.autotune_overlap(lambda: st.step(*mb), iters=1)
        OUT["autotune"] = st.overlap_timings
        restore(st)
        st.step(*mb)
        same_step(st.stats(), ref_stats, "after autotune")
    else:
        # (iv) capture of ONE step with its collective (serial all-reduce: the overlapped form issues async collectives from a hook
        # and refuses capture by design) into a hipGraph -- what SISS_GRAPH_DP=1 asks bench.py to do.  ATTEMPT: the outcome is recorded
        kwt = dict(sample_size=32, in_channels=3, out_channels=3, block_out_channels=(64, 128),
                   down_block_types=("DownBlock2D", "AttnDownBlock2D"), up_block_types=("AttnUpBlock2D", "UpBlock2D"),
                   layers_per_block=1, attention_head_dim=None, norm_num_groups=32, norm_eps=1e-6,
                   downsample_padding=0, flip_sin_to_cos=False, freq_shift=1)
        B = 4
        eng = UNetEngine(UNet2DConfig(**kwt), dev)
        sd = eng.init_random(seed=3)
        mb = batch(B, 32, g, dev)
        ref = SISSStepper(eng, ac, train_batch_size=B, process_group=None, **kw)
        ref.step(*mb)
        ref_stats = ref.stats()
        eng.load_state_dict(sd)
        st = SISSStepper(eng, ac, train_batch_size=B, process_group=pg, **kw)
        st.set_overlap(False, "allreduce")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        rec = {"attempted": True}
        try:
            with torch.cuda.stream(side):
                st.step(*mb)                                         # settle allocations + communicator channels on the capture stream
                eng.load_state_dict(sd)
                st.opt.m.zero_(); st.opt.v.zero_(); st.opt.scalars.zero_()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    st.step(*mb)
            torch.cuda.current_stream().wait_stream(side)
            rec["captured"] = True
            eng.load_state_dict(sd)
            st.opt.m.zero_(); st.opt.v.zero_(); st.opt.scalars.zero_()
            graph.replay()
            torch.cuda.synchronize()
            got = st.stats()
            same_step(got, ref_stats, "graph replay")
            rec["replayed"], rec["replay_matches_no_group_step"] = True, True
            rec.update({k: got[k] for k in KEYS})
        except Exception as e:                                       # recorded, not fatal: the question was whether it works
            rec["captured"] = rec.get("captured", False)
            rec["error"] = f"{type(e).__name__}: {str(e)[:400]}"
        OUT["graph_capture"] = rec
        OUT["no_group"] = {k: ref_stats[k] for k in KEYS}
        # (v) the same as bench.py's autotune candidate (SISSStepper.try_captured_serial, round 6): capture, agree, time, RESTORE --
        # the step after it must equal the no-group step again (parameters, moments and step counter put back)
        if rec.get("replayed"):
            eng.load_state_dict(sd)
            st.opt.m.zero_(); st.opt.v.zero_(); st.opt.scalars.zero_()
            st.set_overlap(True, "allreduce")
            g2, secs, err = st.try_captured_serial(lambda: st.step(*mb))
            cand = {"captured": g2 is not None, "ms_per_step": None if secs is None else secs * 1e3, "error": err,
                    "mode_after": ("overlap" if st.overlap else "serial") + "/" + st.exchange}
            if g2 is not None:
                g2.replay()                                          # the kept graph replays from the restored state
                torch.cuda.synchronize()
                same_step(st.stats(), ref_stats, "replay after try_captured_serial")
                cand["state_restored"] = True
            OUT["captured_serial_candidate"] = cand
    finish(True)
    if MODE == "graph":          # a refused capture can leave the communicator in a state whose teardown blocks: the record is out, leave
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    try:
        dist.destroy_process_group()
    except Exception:
        pass


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:
        OUT["error"] = f"{type(e).__name__}: {str(e)[:1500]}"
        finish(False)
        raise
