#!/usr/bin/env python
"""SISS unlearning-step benchmark (BASELINE.json metric: unlearning-steps/sec + samples/sec,
CelebA-HQ 256x256 DDPM, SISS lambd=0.5, bf16, bs=16 per GPU, 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one optimizer update of delete_celeb.py:557-773 with gradient accumulation 1:
fused mixture/IS-weight kernel -> UNet forward -> fused loss-seed kernel -> dual-cotangent UNet
backward -> (N>1: ONE all-reduce of [g_x ; g_a]) -> norm-fix + clip + AdamW.  Inputs are synthetic
(x0 ~ U[-1,1], a0 = one image repeated, noise ~ N(0,1), t = 999, u ~ U[0,1)), resident in HBM
before the timed region; weights are random-init at the exact google/ddpm-celebahq-256 shapes.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_SAMPLE = 498.35          # SURVEY.md §8d (2*MAC: conv + linear + attention matmuls + GN)
SD_FWD_GFLOP_PER_SAMPLE = 803.9        # SD v1 UNet at 64x64 latents, 77 text tokens
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0                 # HBM3E (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (BASELINE: 16)")
    ap.add_argument("--grad-accum", type=int, default=1,
                    help="micro-batches per optimizer step (config/delete_celeb.yaml ships 16 x batch 4; BASELINE: 1)")
    ap.add_argument("--config", default="celebahq256", choices=["celebahq256", "small", "sd15"],
                    help="celebahq256 = the BASELINE metric's workload (configs[1]); sd15 = BASELINE configs[4] "
                         "(SD v1.5 UNet, 64x64 latents, text conditioning) as a secondary measurement")
    ap.add_argument("--loss-fn", default="importance_sampling_with_mixture")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("SISS_GRAPH", "1")),
                    help="replay the step from a captured hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--engine-attr", action="append", default=[], help="name=int: set a schedule switch of the engine (A/B runs)")
    ap.add_argument("--stepper-attr", action="append", default=[], help="name=int: set a switch of the stepper (A/B runs)")
    ap.add_argument("--lib-set", action="append", default=[], help="name=int: call a process-wide setter of the library (A/B runs)")
    ap.add_argument("--lib", default=None, help="path of another build of libsiss_hip.so (same-box A/B of a kernel change)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


CPU_MICRO_BATCH = 4          # the CPU oracle walks the B samples of a step in micro-batches of 4 (bounded host memory)
CPU_TIMED_STEPS = 3          # 1 warm-up + 3 timed steps, the median is reported


def cpu_baseline(cfg_kw, seed, sd=False, B=16):
    """The oracle (CPU restatement pinned by the reference's golden vectors) timed on the host cores at the STATED
    configuration: ONE optimizer step over the full per-GPU batch B, fp32 -- executed as B / 4
    micro-batches of 4 with gradient accumulation, the way config/delete_celeb.yaml itself ships the step (batch 4 x GA 16;
    same arithmetic: the loss is normalised by B, GroupNorm / attention are per sample), which bounds the autograd memory on
    the host.  One untimed warm-up step at batch 4 first (thread pool, oneDNN primitive caches, first touch of the
    parameters + optimizer state), then CPU_TIMED_STEPS timed steps.  Returns (median seconds for the B-sample step, cores,
    all timed steps' seconds)."""
    from oracle import schedule as S
    from oracle.loss import OracleDeletionLoss
    from oracle.step import unlearning_step
    from oracle.unet import OracleUNet2D, UNetConfig
    from oracle.unet_cond import OracleUNet2DCondition, UNetCondConfig
    # usable cores: the affinity mask (a container may expose far fewer than os.cpu_count()), capped at
    # 32 threads -- beyond that torch's CPU convolutions stop scaling and oversubscription can stall.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:                                   # cgroup v2 CPU quota ("max" or "<quota> <period>")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = min(avail, max(1, int(q) // int(per)))
    except Exception:
        pass
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    net = OracleUNet2DCondition(UNetCondConfig(**cfg_kw)) if sd else OracleUNet2D(UNetConfig(**cfg_kw))
    g = torch.Generator().manual_seed(seed)
    hw = cfg_kw["sample_size"]
    c = cfg_kw["in_channels"]
    x0 = torch.rand(B, c, hw, hw, generator=g) * 2 - 1
    a0 = (torch.rand(1, c, hw, hw, generator=g) * 2 - 1).repeat(B, 1, 1, 1)
    noise = torch.randn(B, c, hw, hw, generator=g)
    t = torch.full((B,), 999, dtype=torch.long)
    u = torch.rand(B, generator=g)
    ctx = None
    if sd:
        ac = S.alphas_cumprod(beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012)
        x0, a0 = 0.18215 * x0, 0.18215 * a0
        ctx = torch.randn(1, 77, cfg_kw["cross_attention_dim"], generator=g)
    else:
        ac = S.alphas_cumprod()
    opt = torch.optim.AdamW(net.parameters(), lr=5e-6, betas=(0.95, 0.999), weight_decay=1e-6)
    L = OracleDeletionLoss(*S.gamma_sigma(ac))

    def step(n, micro):
        """one optimizer step over the first n samples in micro-batches of `micro` (train_batch_size x GA = n)"""
        mbs = [dict(x0=x0[i:i + micro], a0=a0[i:i + micro], noise=noise[i:i + micro], t=t[i:i + micro], u=u[i:i + micro])
               for i in range(0, n, micro)]
        cond = {"encoder_hidden_states": ctx.repeat(micro, 1, 1)} if sd else None
        t0 = time.perf_counter()
        unlearning_step(net, opt, L, "importance_sampling_with_mixture", ac, mbs, train_batch_size=micro, scaling_norm=500.0,
                        loss_params={"lambd": 0.5}, conditioning=cond)
        return time.perf_counter() - t0
    micro = min(CPU_MICRO_BATCH, B)
    step(micro, micro)                                              # warm-up, untimed (1 warm-up + 3 timed)
    n = B - B % micro if B >= micro else B
    dts = [step(n, micro) for _ in range(CPU_TIMED_STEPS)]
    return sorted(dts)[len(dts) // 2], cores, dts


def hbm_traffic(launcher):
    """Average HBM bytes per launch of `launcher`'s kernels from the last committed PMC run
    (profiles/latest_hbm_traffic.json, written by tools/pmc_traffic.sh: 2 x FETCH_SIZE + WRITE_SIZE in separate
    --pmc passes).  PMC counters cannot be read from inside this process, so this is a recorded value of the
    same command, or None when no profile is committed.  Returns (bytes per launch, provenance): the profile carries a
    fingerprint of the kernel sources it was taken on; a mismatch with the sources of THIS build is reported (and warned
    about), so a stale profile cannot pass for a measurement of the running kernels."""
    fn = os.path.join(ROOT, "profiles", "latest_hbm_traffic.json")
    if not os.path.exists(fn):
        return None, "no committed profile"
    prof = json.load(open(fn))
    meta = prof.pop("__meta__", {})
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from parse_pmc import csrc_fingerprint
    now = csrc_fingerprint(ROOT)
    fresh = meta.get("csrc_sha16") == now
    src = ("profiles/latest_hbm_traffic.json (tag %s): 2 x FETCH_SIZE + WRITE_SIZE of this kernel from two separate rocprofv3 --pmc "
           "passes of the same command (tools/pmc_traffic.sh); kernel sources of the profile %s, of this build %s: %s"
           % (meta.get("tag", "?"), meta.get("csrc_sha16", "unrecorded"), now,
              "SAME sources" if fresh else "STALE -- re-run tools/pmc_traffic.sh"))
    if not fresh:
        print("[bench] warning: profiles/latest_hbm_traffic.json was not taken on this build's kernel sources", file=sys.stderr)
    stem = launcher.replace("siss_", "")              # a kernel symbol (gemm_nt_c3p_kernel) or a launcher's family
    tot = n = 0.0
    for k, v in prof.items():
        if k.startswith(stem):
            tot += v["hbm_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    return (round(tot / n) if n else None), src


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process -- before this process has
    made any GPU call (it never makes one) -- let it inherit stdout / stderr, so rank 0's JSON line is this command's JSON line, and
    return its exit code.  What `accelerate launch` + `accelerator.prepare` do for the reference (delete_celeb.py:99-101,304)."""
    import socket
    import subprocess
    with socket.socket() as s:                      # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if os.environ.get("SISS_BENCH_SINGLE_DEVICE") == "1":   # smoke-testing the N>1 code path on a 1-GPU box (gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SISS_DIST_BACKEND", "nccl")           # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    from siss_amd import lib
    from siss_amd.config import UNet2DConfig
    from siss_amd.step import SISSStepper
    from siss_amd.unet import UNetEngine
    if a.lib:
        lib.LIB_PATH = os.path.abspath(a.lib)

    sd = a.config == "sd15"
    if a.config == "celebahq256":
        cfg = UNet2DConfig.celebahq256()
    elif sd:
        from siss_amd.config import UNet2DConditionConfig
        from siss_amd.unet_cond import UNetCondEngine
        cfg = UNet2DConditionConfig.sd15()
    else:
        cfg = UNet2DConfig(sample_size=64, block_out_channels=(128, 128, 256),
                           down_block_types=("DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D"))
    B, hw, cin = a.batch, cfg.sample_size, cfg.in_channels
    eng = UNetCondEngine(cfg, dev) if sd else UNetEngine(cfg, dev)
    for kv in filter(None, a.lib_set):
        k, v = kv.split("=")
        lib.query(k, int(v))
    for kv in filter(None, a.engine_attr):                   # A/B of a schedule switch on one box, e.g. --engine-attr fold_shortcut=0
        k, v = kv.split("=")
        assert hasattr(eng, k), k
        setattr(eng, k, type(getattr(eng, k))(int(v)))
    eng.init_random(seed=42)                       # same weights on every rank (config/delete_celeb.yaml:4)
    g = torch.Generator(device=dev).manual_seed(42 + rank)      # per-rank shard of the synthetic stream
    cond = None
    if sd:
        # delete_sd.yaml: scaled-linear betas, AdamW lr 1e-5 / (0.9, 0.999) / wd 1e-2, scaling_norm 750; latents carry
        # the VAE scaling factor 0.18215 (delete_sd.py:883,888); ONE prompt embedding repeated (delete_sd.py:941-944)
        ac = torch.cumprod(1.0 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2, 0)
        st = SISSStepper(eng, ac, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, scaling_norm=750.0,
                         lambd=0.5, train_batch_size=B, grad_accum=a.grad_accum, loss_fn=a.loss_fn, process_group=pg,
                         mixed_precision="bf16")
        x0 = (0.18215 * torch.randn(B, cin, hw, hw, generator=g, device=dev)).to(torch.bfloat16)
        a0 = (0.18215 * torch.randn(1, cin, hw, hw, generator=g, device=dev)).repeat(B, 1, 1, 1).to(torch.bfloat16)
        cond = {"encoder_hidden_states": torch.randn(1, 77, cfg.cross_attention_dim, generator=g, device=dev)
                .repeat(B, 1, 1).to(torch.bfloat16)}
    else:
        ac = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32), 0)
        st = SISSStepper(eng, ac, lr=5e-6, betas=(0.95, 0.999), eps=1e-8, weight_decay=1e-6,   # delete_celeb.yaml:127-133
                         scaling_norm=500.0, lambd=0.5, train_batch_size=B, grad_accum=a.grad_accum, loss_fn=a.loss_fn,
                         process_group=pg, mixed_precision="bf16")
        x0 = (torch.rand(B, cin, hw, hw, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
        a0 = (torch.rand(1, cin, hw, hw, generator=g, device=dev) * 2 - 1).repeat(B, 1, 1, 1).to(torch.bfloat16)
    for kv in filter(None, a.stepper_attr):
        k, v = kv.split("=")
        assert hasattr(st, k), k
        setattr(st, k, type(getattr(st, k))(int(v)))
    # The per-micro-step draws of the reference loop are INSIDE the timed step (delete_celeb.py:581 noise =
    # randn(shape, dtype=weight_dtype), :593 t = randint(999, 1000), ddpm_deletion_loss.py:18 rand(B) > lambd): device
    # RNG from the default generator, whose philox offset advances correctly under hipGraph replay.
    # (a stream of its own: seeded like `g` above, the default generator's first draw would BE the normals x0 was made from -- the
    # SD line's first step then had noise = x0 / 0.18215, every importance weight saturated to (2, 0), ||g_a|| = 0 and an infinite
    # scaling factor: NaN step scalars in rounds 2-5's SD lines; the kernels' times do not depend on the values)
    torch.cuda.manual_seed(1234 + rank)
    t_low = 999

    def one_step():
        for _ in range(a.grad_accum):              # the same resident images GA times: one optimizer update
            noise = torch.randn(B, cin, hw, hw, device=dev, dtype=torch.bfloat16)
            t = torch.randint(t_low, 1000, (B,), device=dev)
            u = torch.rand(B, device=dev)
            st.micro_step(x0, a0, noise, t, u, cond)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(a.warmup, 1)):
        one_step()
    sync()
    if world > 1 and os.environ.get("SISS_DP_AUTOTUNE", "1") == "1":
        st.autotune_overlap(one_step)          # overlapped vs serial gradient exchange: keep the faster (untimed)
        sync()
    graph = None
    graph_note = None
    # Multi-GPU: the step contains an RCCL all-reduce.  A world-size-1 communicator captured and replayed such a step
    # (tests/test_hip_rccl.py); whether the multi-rank kernels do, and whether it pays, is measured here like the exchange modes:
    # "serial + hipGraph" is one more autotune candidate (SISSStepper.try_captured_serial: every rank captures, the ranks agree on the
    # outcome, a failure anywhere falls back to the eager schedule in process).  SISS_GRAPH_DP=0 skips the candidate, =1 takes it
    # whenever it captures.
    gdp = os.environ.get("SISS_GRAPH_DP", "auto")
    # (RCCL only: the gloo rehearsal's collectives synchronise with the host and cannot be captured)
    if world > 1 and a.graph and gdp != "0" and torch.distributed.get_backend() == "nccl":
        prev = (st.overlap, st.exchange)
        g_, secs, err = st.try_captured_serial(one_step)
        tim = getattr(st, "overlap_timings", None) or {}
        eager_best = min([v for k, v in tim.items() if k.endswith("_ms") and isinstance(v, float)], default=None)
        if g_ is not None:
            tim["serial_hipgraph_ms"] = secs * 1e3
            if gdp == "1" or eager_best is None or secs * 1e3 < eager_best:
                graph, graph_note = g_, "serial all-reduce inside the step's hipGraph"
            else:
                st.set_overlap(*prev)                    # an eager mode was faster on this node
                del g_
        else:
            tim["serial_hipgraph_error"] = err
        st.overlap_timings = tim
        sync()
    use_graph = (a.graph and world == 1) or graph is not None
    if use_graph and graph is None:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            one_step()                              # settle allocations on the capture stream
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        sync()
    run = graph.replay if graph is not None else one_step

    # one event per step boundary on the launch stream (~1 us each): the per-step distribution SURVEY.md §8d asks for
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    sync()
    t0 = time.perf_counter()
    host_s = 0.0
    for i in range(a.steps):
        marks[i].record()
        h0 = time.perf_counter()
        run()
        host_s += time.perf_counter() - h0
    marks[a.steps].record()
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 3)
    step_ms = {"p10": pct(0.1), "p50": pct(0.5), "p90": pct(0.9), "source": "HIP events between consecutive steps, rank 0"}
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / a.steps * 1e3
    stats = st.stats()

    # ---- the same step as the shipped task loop runs it (siss_amd/tasks.py: eager launches, no hipGraph, and the
    #      device-to-host copy of every optimizer step's scalars that feeds the log line, read one step later) -- secondary figure ----
    eager_ms = None
    if graph is not None:
        sync()
        t0 = time.perf_counter()
        pending = None
        for _ in range(min(a.steps, 5)):
            one_step()
            handle = st.stats_async()                 # (the task loop reads a step's scalars once the next step is queued)
            if pending is not None:
                pending.get()
            pending = handle
        pending.get()
        sync()
        eager_ms = (time.perf_counter() - t0) / min(a.steps, 5) * 1e3

    # ---- what the chip does while the step runs (untimed leg, rank 0, N = 1): the step sits at the board's power limit, and what a box's
    #      silicon clocks there is most of the box-to-box spread of `value`.  One rocm-smi reading taken WHILE ~1.5 s
    #      of replays are in flight; absent (None) where the tool or the permission is missing ----
    device_state = None
    if rank == 0 and world == 1 and graph is not None and not a.no_kernel_timing:
        device_state = _smi_under_load(lambda: [graph.replay() for _ in range(max(8, int(1500 / max(ms, 1.0))))])
        sync()

    # ---- N > 1 self-check: every rank must hold bit-identical parameters after the timed steps (the replicated
    #      norm-fix / clip / AdamW only stays in step if the exchange really summed [g_x ; g_a] over all ranks) ----
    selfcheck = None
    if world > 1:
        flat = eng.ps.flat
        chk = torch.stack([flat.double().sum(), flat.double().abs().sum(),
                           flat.view(torch.int32).to(torch.int64).sum().double()])
        allchk = [torch.empty_like(chk) for _ in range(world)]
        torch.distributed.all_gather(allchk, chk)
        same = all(torch.equal(allchk[0], c) for c in allchk)
        finite = bool(torch.isfinite(chk).all())
        selfcheck = {"rccl_ranks_seen": len(allchk), "replicas_identical": bool(same), "finite": finite,
                     "backend": torch.distributed.get_backend()}
        assert same and finite, f"data-parallel replicas diverged or went non-finite: {[c.tolist() for c in allchk]}"

    # ---- N > 1: the exchange on its own (untimed leg): one all-reduce of the flat [g_x ; g_a] buffer, SURVEY.md §8e ----
    exchange = None
    if world > 1:
        from siss_amd.dp import EXCHANGES
        g = eng.ps.grads
        nbytes = g.numel() * g.element_size()
        exchange = {"payload_bytes": nbytes,
                    "note": "bus = 2 (N-1)/N x payload / time; per link = bus / min(N-1, 7) xGMI links per GPU; "
                            "allreduce = RCCL all-reduce of the whole flat pair (siss_amd/dp.py)"}
        timings = getattr(st, "overlap_timings", None) or {}
        for name, fn in EXCHANGES.items():
            if fn is None:
                continue                                  # "sharded" is a whole update, not an exchange of the flat pair
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.fill_(1e-3)
            fn(g, pg)                                     # warm the communicator
            sync()
            reps = 5
            e0.record()
            for _ in range(reps):
                fn(g, pg)
            e1.record()
            sync()
            tt = torch.tensor([e0.elapsed_time(e1) / reps], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            ms_x = float(tt.item())
            bus = 2 * (world - 1) / world * nbytes / (ms_x * 1e-3) / 1e9
            exchange[name] = {"ms": round(ms_x, 3), "alg_GBps": round(nbytes / (ms_x * 1e-3) / 1e9, 1),
                              "bus_GBps": round(bus, 1), "bus_GBps_per_link": round(bus / min(world - 1, 7), 1)}
        eng.zero_grad()

    # ---- per-kernel timing of the dominant kernels (eager, HIP events on the launch stream) ----
    roof = None
    kern = {}
    if not a.no_kernel_timing:
        # every rank runs these steps (the step holds a collective when N>1); only rank 0 records events
        lib.PROF = [] if rank == 0 else None
        ksteps = min(a.steps, 3)
        for _ in range(ksteps):
            one_step()
        sync()
        prof, lib.PROF = lib.PROF, None
        # ... and the same kernel with the chip to itself: the weight gradients of the timed schedule run on a side stream beside part
        # of these launches (UNetEngine.wgrad_side), so their event times include what they lose to it; a second eager leg on the
        # ONE-STREAM schedule (same kernels, same shapes) prices the kernel itself
        prof1 = None
        if getattr(eng, "wgrad_side", False):
            saved = (eng.wgrad_side, eng.prep_side)
            eng.wgrad_side, eng.prep_side = False, False
            one_step()
            sync()
            lib.PROF = [] if rank == 0 else None
            for _ in range(ksteps):
                one_step()
            sync()
            prof1, lib.PROF = lib.PROF, None
            eng.wgrad_side, eng.prep_side = saved
    if not a.no_kernel_timing and rank == 0:
        ksym = {}
        hbm = {}
        dom_split = {}
        dom_kind = {}
        for name, s, e, work, shape, sym, nbytes in prof:
            dt_ms = s.elapsed_time(e)
            if sym == "gemm_nt_c3p_kernel":      # the same kernel on the large grids it was built for vs the mid-size layers
                sh = dict(zip(shape[::2], shape[1::2]))
                big = -(-sh["M"] // 128) * -(-sh["N"] // 128) >= 2048
                b = dom_split.setdefault("grids >= 2048 tiles" if big else "grids < 2048 tiles", [0, 0.0, 0.0])
                b[0] += 1; b[1] += dt_ms; b[2] += work
                # ... and the launches that carry a resnet's 1x1 shortcut (forward: extra K-groups; dgrad: extra column tiles -- an
                # HBM-bound product hidden in the launch) apart from the plain 3x3 ones
                kind = ("3x3 + folded 1x1 shortcut dgrad (x tiles)" if "+1x1 N" in sh else
                        "3x3 + folded 1x1 shortcut fprop (K-groups)" if "+1x1 K" in sh else "plain 3x3")
                b = dom_kind.setdefault(kind, [0, 0.0, 0.0])
                b[0] += 1; b[1] += dt_ms; b[2] += work
            if nbytes:
                h = hbm.setdefault(name, [0, 0.0, 0.0])
                h[0] += 1; h[1] += dt_ms; h[2] += nbytes
            d = kern.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1; d[1] += dt_ms; d[2] += work
            if work:
                k = ksym.setdefault(sym, [0, 0.0, 0.0])
                k[0] += 1; k[1] += dt_ms; k[2] += work
        tot_ms = sum(v[1] for v in kern.values())
        # the dominant KERNEL, under the symbol rocprofv3 --kernel-trace --stats lists it by (profiles/): achieved =
        # algorithmic flops of its launches / their summed duration; avg_launch_us is comparable with the CSV's AverageNs
        dom = max(ksym, key=lambda k: ksym[k][1])
        n, tms, work = ksym[dom]
        ach = work / (tms * 1e-3) / 1e12
        traffic, traffic_src = hbm_traffic(dom)
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic if a.config == "celebahq256" else None,     # HBM bytes per launch
                "traffic_source": traffic_src,
                "launches_per_step": n // ksteps, "avg_launch_us": round(tms / n * 1e3, 2),
                "tflop_per_launch": round(work / n / 1e12, 4),
                **({"by_grid": {k: {"achieved": round(v[2] / (v[1] * 1e-3) / 1e12, 2), "launches_per_step": v[0] // ksteps,
                                    "avg_launch_us": round(v[1] / v[0] * 1e3, 2)} for k, v in sorted(dom_split.items())}}
                   if dom == "gemm_nt_c3p_kernel" and dom_split else {}),
                **({"by_kind": {k: {"achieved": round(v[2] / (v[1] * 1e-3) / 1e12, 2), "launches_per_step": v[0] // ksteps,
                                    "avg_launch_us": round(v[1] / v[0] * 1e3, 2)} for k, v in sorted(dom_kind.items())}}
                   if dom == "gemm_nt_c3p_kernel" and len(dom_kind) > 1 else {}),
                "share_of_step_kernel_time": round(tms / tot_ms, 3),
                **({"one_stream_schedule": (lambda w_, t_, n_: {"achieved": round(w_ / (t_ * 1e-3) / 1e12, 2),
                                                                 "frac": round(w_ / (t_ * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                                                 "avg_launch_us": round(t_ / n_ * 1e3, 2),
                                                                 "note": "the same launches with the side stream off (nothing shares the chip)"})(
                        sum(r[3] for r in prof1 if r[5] == dom), sum(r[1].elapsed_time(r[2]) for r in prof1 if r[5] == dom),
                        max(1, sum(1 for r in prof1 if r[5] == dom)))} if prof1 else {}),
                # the HBM-bound launchers of the step (K1, K5, K10-12), each against the HBM peak:
                # ALGORITHMIC bytes (operands read once, results written once; lib.hbm_bytes) / summed launch time
                "hbm_kernels": {k: {"achieved": round(v[2] / (v[1] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                    "frac": round(v[2] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 3),
                                    "launches_per_step": v[0] // ksteps, "ms_per_step": round(v[1] / ksteps, 3)}
                                for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])},
                "all_mfma_kernels": {k: {"achieved": round(v[2] / (v[1] * 1e-3) / 1e12, 2), "launches_per_step": v[0] // ksteps,
                                         "avg_launch_us": round(v[1] / v[0] * 1e3, 2),
                                         "share_of_step_kernel_time": round(v[1] / tot_ms, 3)}
                                     for k, v in sorted(ksym.items(), key=lambda kv: -kv[1][1])}}

    cpu = None
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        cfg_kw = {k: getattr(cfg, k) for k in ("sample_size", "in_channels", "out_channels", "block_out_channels",
                                                "down_block_types", "up_block_types", "layers_per_block",
                                                "attention_head_dim", "norm_num_groups", "norm_eps",
                                                "downsample_padding", "flip_sin_to_cos", "freq_shift")
                  + (("cross_attention_dim",) if sd else ())}
        cdt, cores, cdts = cpu_baseline(cfg_kw, 42, sd, B)
        try:
            model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:
            model = None
        Bc = B - B % min(CPU_MICRO_BATCH, B) if B >= CPU_MICRO_BATCH else B
        cpu = {"value": round(Bc / cdt, 5), "unit": "samples/sec", "cores": cores, "cpu_model": model,
               "kind": "port",
               "sample": f"1 untimed warm-up step at batch {min(CPU_MICRO_BATCH, B)}, then {len(cdts)} timed optimizer steps (median reported) "
                         f"over the full per-GPU batch of {Bc} (as {max(Bc // CPU_MICRO_BATCH, 1)} micro-batches of {min(CPU_MICRO_BATCH, B)} with "
                         f"gradient accumulation: same arithmetic, bounded host memory) of the same UNet / resolution, fp32 torch CPU oracle, "
                         f"{cdt:.1f} s per step",
               "extrapolated": False,
               "steps_per_sec_at_bs%d" % Bc: round(1.0 / cdt, 6),
               "timed_steps_s": [round(x, 2) for x in cdts]}

    if rank == 0 and cpu is None and world > 1:
        # N > 1: the CPU baseline is timed at N = 1 only (rank 0, one GPU's workload); a scaling record carries that measurement BY VALUE
        # from the committed N = 1 line of this workload, so that it is self-contained
        try:
            ref_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                    "latest_bench_sd15_bs%d.json" % B if sd else "latest_bench_celeb_bs%d.json" % B)
            ref = json.load(open(ref_file))
            if ref.get("cpu_baseline"):
                cpu = dict(ref["cpu_baseline"], measured_at="n_gpus = 1 (per-GPU workload; not re-timed in this run)",
                           source="profiles/" + os.path.basename(ref_file))
        except Exception:
            cpu = None
    if rank == 0:
        steps_per_sec = 1e3 / ms
        GA = a.grad_accum
        step_tflop = 5 * FWD_GFLOP_PER_SAMPLE * B * GA / 1e3 if a.config == "celebahq256" else (
            5 * SD_FWD_GFLOP_PER_SAMPLE * B * GA / 1e3 if sd else None)
        workload = {"celebahq256": "delete_celeb.yaml: CelebA-HQ 256x256 DDPM UNet (113.7M params), SISS lambd=0.5, "
                                   "t=999, bf16, bs=%d/GPU, GA=%d, scaling_norm=500, AdamW lr 5e-6" % (B, GA),
                    "sd15": "delete_sd.yaml: Stable Diffusion v1.5 UNet (859.5M params), 64x64x4 latents + 77x768 text "
                            "embedding, SISS lambd=0.5, t=999, bf16, bs=%d/GPU, GA=%d, scaling_norm=750, AdamW lr 1e-5, "
                            "no gradient checkpointing (activations kept in HBM)" % (B, GA),
                    "small": "small 64x64 dev config"}[a.config]
        out = {
            "metric": "unlearning samples/sec (= unlearning-steps/sec x batch x gpus), "
                      + ("SD-v1.5 UNet SISS (secondary; BASELINE metric is the CelebA-HQ line)" if sd else "CelebA-HQ-256 DDPM SISS"),
            "value": round(steps_per_sec * B * GA * world, 3), "unit": "samples/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "steps_per_sec": round(steps_per_sec, 4), "step_ms": step_ms,
            # host time inside the launch calls of the timed steps (hipGraphLaunch returns when its packets are queued).  The two-stream
            # graph takes ~28-40 us per kernel node here (33 of SD B = 4's 43 ms; a linear graph 0.3 ms) and still stays ahead of the
            # device: the linear schedule measures the same step time (docs/experiments.md, round 6)
            "launch_host_ms_per_step": round(host_s / a.steps * 1e3, 3),
            # the gradient fill of a step: bytes of the [g_x ; g_a] buffer, and how many of them the fill skips because the backward pass
            # overwrites them (UNetEngine.zero_grad(sparse_key=...); None: full fill)
            "gradient_fill": {"buffer_bytes": eng.ps.grads.numel() * 4,
                              "skipped_bytes": max([p["skipped_bytes"] for p in getattr(eng, "_fill_plans", {}).values()], default=None)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": workload,
                       "loss_fn": a.loss_fn, "global_batch": B * GA * world, "grad_accum": GA, "parallelism": f"dp{world}",
                       "hipgraph": bool(use_graph), "rng_in_timed_region": True,
                       "eager_task_loop_ms_per_step": round(eager_ms, 3) if eager_ms else None,
                       **({"dp_selfcheck": selfcheck} if selfcheck else {}),
                       **({"dp_exchange": graph_note or ("overlapped all-reduce (two grouped collectives)" if st.overlap else "serial " + st.exchange),
                           "dp_autotune": getattr(st, "overlap_timings", None),
                           "dp_allreduce": exchange} if world > 1 else {})},
            "step_tflop_algorithmic": step_tflop,
            "step_mfma_frac": round(step_tflop / (ms * 1e-3) / PEAK_BF16_TFLOPS, 4) if step_tflop else None,
            "roofline": roof, "cpu_baseline": cpu,
            "step_scalars": {k: stats[k] for k in ("norm_loss_x", "norm_loss_a", "scaling_factor", "pre_clip_norm")},
            **({"device_under_load": device_state} if device_state else {}),
            "kernel_ms_per_step": {k: round(v[1] / max(min(a.steps, 3), 1), 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])[:12]},
        }
        print(json.dumps(out))
        if world > 1:
            # one grep-able line beside the JSON (stderr: stdout carries the ONE JSON line): did every rank take part, which exchange
            # ran, what a link moved
            ar = (exchange or {}).get("allreduce") or {}
            print(f"[siss dp] rccl_ranks_seen={(selfcheck or {}).get('rccl_ranks_seen')} world={world} "
                  f"mode={out['config']['dp_exchange']!r} hipgraph={bool(use_graph)} "
                  f"allreduce_ms={ar.get('ms')} bus_GBps_per_link={ar.get('bus_GBps_per_link')} ms_per_step={out['ms_per_step']}",
                  file=sys.stderr, flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def _smi_under_load(enqueue):
    """Power / shader clock / temperature from ONE `rocm-smi` call made while the work `enqueue()` queued is running (the replays
    are asynchronous: the call overlaps them).  None when rocm-smi is not there or fails."""
    import re
    import shutil
    import subprocess
    # Never under a profiler: rocprofv3's preloaded library initialises the GPU in every child that inherits its environment, and
    # rocm-smi is a `#!/usr/bin/env python3` script -- the env -> python3 hop would then be an exec of a GPU-initialised process
    # (ADVICE r05).  A profiler shows as ROCP_* / ROCPROFILER_* / ROCPROF_* / HSA_TOOLS_* variables or a rocprof library in LD_PRELOAD
    # (LD_PRELOAD alone does not: the GPU boxes of this pool preload a guard of their own in every process).  Otherwise: the
    # interpreter invoked directly on the script (no shebang hop), with any tool variables scrubbed from the child's environment.
    tool_keys = ("ROCP_", "ROCPROFILER_", "ROCPROF_", "HSA_TOOLS_")
    if any(k.startswith(tool_keys) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None
    script = "/opt/rocm/libexec/rocm_smi/rocm_smi.py"
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if os.path.exists(script):
        cmd = [sys.executable, script]
    elif os.path.exists(exe):
        cmd = [exe]
    else:
        return None
    env = {k: v for k, v in os.environ.items() if not k.startswith(tool_keys)}
    try:
        enqueue()
        time.sleep(0.3)                                # (the first replays ramp the clocks)
        r = subprocess.run(cmd + ["--showpower", "--showmaxpower", "--showclocks", "--showtemp"], capture_output=True, text=True,
                           timeout=20, env=env)
        txt = r.stdout
        grab = lambda pat: (lambda m: float(m.group(1)) if m else None)(re.search(pat, txt))
        out = {"power_w": grab(r"Current Socket Graphics Package Power \(W\): ([0-9.]+)"),
               "power_cap_w": grab(r"Max Graphics Package Power \(W\): ([0-9.]+)"),
               "sclk_mhz": grab(r"sclk clock level: \S+ \(([0-9.]+)Mhz\)"),
               "mclk_mhz": grab(r"mclk clock level: \S+ \(([0-9.]+)Mhz\)"),
               "junction_c": grab(r"Sensor junction\) \(C\): ([0-9.]+)"),
               "source": "one rocm-smi reading while ~1.5 s of hipGraph replays of the step were in flight (untimed leg)"}
        return out if out["power_w"] is not None else None
    except Exception:
        return None


if __name__ == "__main__":
    main()
