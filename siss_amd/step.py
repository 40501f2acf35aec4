"""One SISS unlearning optimizer step on HIP -- the hot loop body of the reference's
delete_celeb.py:557-773 (delete_tshirt.py:501-717), in closed form:

    per micro-batch k:  x_mix, iw     = mixture_fwd(x0, a0, noise, t, u)                   [fused pre-kernel]
                        pred          = UNet(x_mix, t)                                      [one forward]
                        c_x, c_a      = d/dpred sum(iw * (pred - eps)^2) / (B_total * GA)   [fused post-kernel]
                        [g_x ; g_a]  += J^T [c_x ; c_a]                                    [ONE dual-cotangent backward]
    sync step:          (all-reduce [g_x ; g_a] over ranks)                                 [one RCCL collective]
                        s = scaling_norm / ||g_a|| ; g = g_x - s g_a ; clip ; AdamW         [two flat passes]

What the reference does with two ``backward`` calls, three 450-tensor clone loops, two norm
loops and a foreach optimizer is here a fixed kernel schedule with no host synchronisation;
the logged scalars stay on the device until ``stats()`` is called.

Data parallelism: every rank runs the same schedule on its own keep / forget
shard; the only exchange is one sum all-reduce of the flat [g_x ; g_a] buffer per optimizer
step.  Loss normaliser = global batch (train_batch_size * world_size), so the result equals
the single-process step on the concatenated batch.
"""
import torch

from . import lib
from . import dp
from .dp import EXCHANGES, allreduce_flat_grads, allreduce_pieces, can_shard
from .loss import loss_bwd_seed, mixture_fwd, mse_bwd_seed
from .optim import FlatAdamW
from .unet import UNetEngine

SISS = "importance_sampling_with_mixture"
NO_IS = "double_forward_with_neg_del"
ERASEDIFF = "erasediff"
NEG_GRAD = "simple_neg_del"
NAIVE = "naive_del"
SUBSCORE = "subscore_bernoulli"


class SISSStepper:
    _warned_fp32 = False
    wgrad_overwrite = True          # (A/B switch; see UNetEngine.wgrad_overwrite)

    def __init__(self, engine: UNetEngine, alphas_cumprod, *, lr, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-2, scaling_norm=None, eta=None, lambd=0.5, train_batch_size,
                 grad_accum=1, max_grad_norm=1.0, loss_fn=SISS, inf_guard=False, process_group=None,
                 mixed_precision="bf16", superfactor=1.0, superfactor_decay=None):
        self.e = engine
        dev = engine.device
        ac = alphas_cumprod.to(device=dev, dtype=torch.float32).contiguous()
        self.ac = ac
        self.gamma_tab = (ac ** 0.5).contiguous()           # delete_celeb.py:367-371
        self.sigma_tab = ((1 - ac) ** 0.5).contiguous()
        self.lambd, self.scaling_norm, self.eta, self.inf_guard = float(lambd), scaling_norm, eta, inf_guard
        self.train_batch_size, self.ga = int(train_batch_size), int(grad_accum)
        self.loss_fn = loss_fn
        self.superfactor = float(superfactor)
        self.superfactor_decay = None if superfactor_decay is None else float(superfactor_decay)   # delete_celeb.py:658-662
        if loss_fn == ERASEDIFF:
            assert eta is not None, "erasediff needs eta (delete_celeb.py:740-742)"
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        # the step runs its collectives: more than one rank (or dp.FORCE_COLLECTIVES, the test-only switch for a 1-rank RCCL group)
        self.dp_on = process_group is not None and (self.world > 1 or dp.FORCE_COLLECTIVES)
        self.io_dtype = torch.bfloat16 if mixed_precision == "bf16" else torch.float32
        f32_engine = getattr(engine, "f32", False)
        if f32_engine and mixed_precision == "bf16":
            raise ValueError("mixed_precision='bf16' on an f32 engine: build the engine with dtype=torch.bfloat16")
        if mixed_precision != "bf16" and not f32_engine and not SISSStepper._warned_fp32:
            # config/delete_*.yaml ship mixed_precision: null (fp32 everywhere in the reference).  On an f32 ENGINE
            # (UNetEngine(dtype=torch.float32): what the task loop builds for null) that is what runs.  On a bf16 engine it selects
            # f32 image / noise I/O, f32 master weights, gradients and optimizer state -- with bf16 MFMA operands.
            import warnings
            warnings.warn("siss_amd: mixed_precision is not 'bf16' on a bf16 engine: I/O, master weights, gradients and optimizer "
                          "state are f32, but convolutions / linears run on bf16 MFMA operands with f32 accumulation -- results "
                          "match an fp32 run to bf16 tolerance; UNetEngine(cfg, device, dtype=torch.float32) is the f32 path",
                          RuntimeWarning, stacklevel=2)
            SISSStepper._warned_fp32 = True
        self.opt = FlatAdamW(engine.ps.flat, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                             max_grad_norm=max_grad_norm, shadow=None if f32_engine else engine.ps.shadow)
        self._micro = 0
        self.last = None
        self._pending = []
        # Data-parallel overlap: the tail [split, P) of the flat gradient buffer (up / mid / deep down blocks,
        # ~85 % of the bytes) is final long before the high-resolution down blocks finish their backward;
        # its all-reduce is started from a hook inside the backward pass and runs beside the rest of it.
        import os
        self.exchange = os.environ.get("SISS_DP_EXCHANGE", "allreduce")      # serial mode: "allreduce" | "sharded"
        self._state_shard = None                # (lo, hi) while AdamW's moments are current on this rank's shard only
        assert self.exchange in EXCHANGES
        self.set_overlap(self.dp_on and engine.ps.split < engine.ps.total
                         and os.environ.get("SISS_DP_OVERLAP", "1") != "0")

    def set_overlap(self, on, exchange=None):
        """Overlapped (tail all-reduce started from inside the backward) or serial (one exchange after it: RCCL's
        all-reduce, or the sharded update).  Moments left on a shard by sharded steps are re-gathered where the next
        replicated update needs them (_sync_and_update), whichever way the mode was changed."""
        self.overlap = bool(on)
        if exchange is not None:
            assert exchange in EXCHANGES
            self.exchange = exchange
        self.e.on_early_grads_final = self._early_allreduce if self.overlap else None

    def _gather_optimizer_state(self):
        """The sharded update advances AdamW's moments on this rank's parameter shard only.  Before a REPLICATED update
        follows (another exchange mode, a checkpoint of the optimizer) the shards are all-gathered once, so that every
        rank holds identical full moments again."""
        if self._state_shard is None:
            return
        from .dp import all_gather_params
        lo, hi = self._state_shard
        all_gather_params(self.opt.m, lo, hi, self.pg)
        all_gather_params(self.opt.v, lo, hi, self.pg)
        self._state_shard = None

    def optimizer_state(self):
        """(m, v, scalars) with the moments complete on every rank -- what a checkpoint of the optimizer saves."""
        self._gather_optimizer_state()
        return self.opt.m, self.opt.v, self.opt.scalars

    def autotune_overlap(self, step_fn, iters=3):
        """Measure, don't guess: the overlapped exchange shares the chip with the persistent one-block-per-CU GEMMs of
        the high-resolution backward, and whether RCCL's workgroups cost those kernels more than the overlap hides
        depends on the node.  Times `iters` steps each way (max over ranks) and keeps the faster setting."""
        if not self.dp_on or self.e.ps.split >= self.e.ps.total:
            return self.overlap
        import time
        dist = torch.distributed
        results = {}
        # the three settings a node can tell apart: one all-reduce after the backward, the same exchange overlapped with it,
        # and the sharded update (reduce-scatter -> shard-local AdamW -> all-gather of the parameters)
        candidates = {"overlap": (True, "allreduce"), "serial": (False, "allreduce")}
        if can_shard(self.e.ps.total, self.world):
            candidates["serial_sharded"] = (False, "sharded")
        # the timed steps are real optimizer steps: put parameters, AdamW moments, the step counter and the stepper's own
        # state (NegGrad's decaying superfactor, the last step's statistics) back afterwards -- the run that follows starts
        # from what it was given, whichever candidate wins
        self._gather_optimizer_state()          # sharded steps before this call left the moments current on the owner's shard only
        saved = [t.clone() for t in (self.opt.p, self.opt.m, self.opt.v, self.opt.scalars)]
        saved_state = (self.superfactor, self.last, self._micro)
        errors = {}
        for name, (mode, exch) in candidates.items():
            self.set_overlap(mode, exch)
            exc = None
            try:
                step_fn()                                        # settle (scratch buffers, communicator channels)
            except (RuntimeError, NotImplementedError) as e:
                exc = e
            # A failure may be rank-LOCAL (torch's OOM is a RuntimeError; a launch error): the ranks AGREE on it before any of
            # them decides -- a rank that skipped the candidate alone would leave the others waiting in the barrier below.
            gathered = [None] * self.world
            dist.all_gather_object(gathered, exc is not None, group=self.pg)
            if any(gathered):
                # collectives a backward hook already started must retire before the next candidate rewrites the gradient
                # buffer they read -- wait on them, do not just forget them
                for work in self._pending:
                    try:
                        if work is not None:
                            work.wait()
                    except Exception:
                        pass
                torch.cuda.synchronize()
                self._pending, self._micro, self._state_shard = [], 0, None
                if name == "serial":            # the serial all-reduce must work or the run has no exchange
                    raise exc if exc is not None else RuntimeError("the serial gradient exchange failed on another rank")
                errors[name] = ((str(exc) or type(exc).__name__) if exc is not None else "failed on another rank")[:200]
                continue
            dist.barrier(group=self.pg); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                step_fn()
            dist.barrier(group=self.pg); torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t0], device=self.e.device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.pg)
            results[name] = float(tt.item()) / iters
        best = min(results, key=results.get)
        self._state_shard = None                                 # (the moments are overwritten as a whole below)
        for dst, src in zip((self.opt.p, self.opt.m, self.opt.v, self.opt.scalars), saved):
            dst.copy_(src)
        del saved
        self.superfactor, self.last, self._micro = saved_state
        self.e.refresh_weights(cast_shadow=True)
        self.set_overlap(best.startswith("overlap"), candidates[best][1])
        self.overlap_timings = {k + "_ms": v * 1e3 for k, v in results.items()}
        if errors:
            self.overlap_timings["errors"] = errors
        return self.overlap

    def try_captured_serial(self, step_fn, iters=3):
        """N > 1 insurance (round 6): the step with its SERIAL all-reduce captured into a hipGraph, as one more candidate beside the eager
        exchange modes (a world-size-1 RCCL communicator captured and replayed such a step: tests/test_hip_rccl.py; whether the
        multi-rank kernels do is the node's to say).  Captures on every rank, AGREES on the outcome (a rank-local failure makes every
        rank fall back: nobody replays a graph the others do not have), times `iters` replays (max over ranks) and restores
        parameters / moments / stepper state.  Returns (graph, seconds per step, None) with the stepper left on the serial exchange,
        or (None, None, reason) with the previous exchange mode restored -- in process, no re-exec, no relaunch."""
        dist = torch.distributed
        import time
        self._gather_optimizer_state()
        saved = [t.clone() for t in (self.opt.p, self.opt.m, self.opt.v, self.opt.scalars)]
        saved_state = (self.superfactor, self.last, self._micro)
        prev = (self.overlap, self.exchange)
        self.set_overlap(False, "allreduce")
        graph, err, secs = None, None, None
        try:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=self.e.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                step_fn()                                            # settle allocations on the capture stream
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    step_fn()
            cur.wait_stream(side)
            graph.replay()
            torch.cuda.synchronize()
        except Exception as e:                                       # capture refused / launch error: eager it is
            err, graph = (str(e) or type(e).__name__)[:200], None
        flags = [None] * self.world
        dist.all_gather_object(flags, err is None, group=self.pg)
        if all(flags):
            dist.barrier(group=self.pg); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                graph.replay()
            dist.barrier(group=self.pg); torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t0], device=self.e.device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.pg)
            secs = float(tt.item()) / iters
        else:
            graph, err = None, err or "capture failed on another rank"
            torch.cuda.synchronize()
        self._pending, self._state_shard = [], None
        for dst, src in zip((self.opt.p, self.opt.m, self.opt.v, self.opt.scalars), saved):
            dst.copy_(src)
        self.superfactor, self.last, self._micro = saved_state
        self.e.refresh_weights(cast_shadow=True)
        if graph is None:
            self.set_overlap(*prev)
        return graph, secs, err

    # ------------------------------------------------------------------ one micro-batch
    def micro_step(self, x0, a0, noise, t, u, conditioning=None, erase_target=None):
        """Inputs: x0/a0/noise [B,C,H,W] (cast to the io dtype like delete_celeb.py:561-581),
        t [B] int64, u [B] keep/forget uniforms; conditioning: the dict the reference splats into the UNet
        call ({} or {'encoder_hidden_states': [B,77,768]}, delete_sd.py:974-976).  Enqueues fwd + dual
        backward; no host sync."""
        e = self.e
        cond = dict(conditioning or {})
        # first micro-batch of a step: one-split weight gradients overwrite their tiles (ONE backward pass per micro-batch below)
        e.wgrad_overwrite = self._micro == 0 and self.wgrad_overwrite
        if self._micro == 0:
            # the fill runs beside the forward pass (joined by the backward pass) and skips what that backward pass will overwrite:
            # the key names everything its split decisions depend on
            key = (self.loss_fn, tuple(x0.shape), tuple(sorted((k, tuple(v.shape)) for k, v in cond.items())))
            e.zero_grad(beside_forward=True, sparse_key=key)
        x0, a0, noise = (v.to(device=e.device, dtype=self.io_dtype).contiguous() for v in (x0, a0, noise))
        B = x0.shape[0]
        scale = 1.0 / (self.train_batch_size * self.world * self.ga)
        if self.loss_fn == SISS:
            m = mixture_fwd(x0, a0, noise, t, u, self.ac, self.gamma_tab, self.sigma_tab, self.lambd)
            pred = e.forward(m.x_mix, t, **cond)
            cot = e._buf("cot", (2 * B, *pred.shape[1:]))
            # c_x / c_a go straight into the stacked cotangent buffer that seeds the dual backward
            seed = loss_bwd_seed(pred, m, x0, a0, scale, c_out=cot, partials=self._partials(B, pred[0].numel()))
            e.backward(cot, nsets=2)
            chw = pred[0].numel()
            self.last = dict(loss_x=seed.sum_loss_x / chw, loss_a=seed.sum_loss_a / chw, iw_x=m.iw_x, iw_a=m.iw_a)
        elif self.loss_fn == NO_IS:
            # two forwards as ONE batch-2B forward (ddpm_deletion_loss.py:60-67), plain noise target
            m = mixture_fwd(x0, a0, noise, t, torch.ones(B, device=e.device), self.ac, self.gamma_tab,
                            self.sigma_tab, 0.0)   # u=1 > 0: keep rows -> q_sample(x0)
            mf = mixture_fwd(x0, a0, noise, t, torch.zeros(B, device=e.device), self.ac, self.gamma_tab,
                             self.sigma_tab, 0.5)  # u=0 <= .5: forget rows -> q_sample(a0)
            xin = torch.cat([m.x_mix, mf.x_mix], 0)
            cond2 = {k: torch.cat([v, v], 0) for k, v in cond.items()}
            pred = e.forward(xin, torch.cat([t, t], 0), **cond2)
            tgt = torch.cat([noise, noise], 0)
            cot, _, sums = mse_bwd_seed(pred, tgt, scale)
            e.backward(cot, nsets=2)
            chw = pred[0].numel()
            self.last = dict(loss_x=sums[:B] / chw, loss_a=sums[B:] / chw)
        elif self.loss_fn == ERASEDIFF:
            # ddpm_deletion_loss.py:70-78: keep half against the noise, forget half against U[0,1) "noise"; the
            # recombination is s = -max(eta - <g_x, g_a> / |g_a|^2, 0) (delete_celeb.py:740-742, optimizer mode)
            m = mixture_fwd(x0, a0, noise, t, torch.ones(B, device=e.device), self.ac, self.gamma_tab,
                            self.sigma_tab, 0.0)
            mf = mixture_fwd(x0, a0, noise, t, torch.zeros(B, device=e.device), self.ac, self.gamma_tab,
                             self.sigma_tab, 0.5)
            xin = torch.cat([m.x_mix, mf.x_mix], 0)
            cond2 = {k: torch.cat([v, v], 0) for k, v in cond.items()}
            pred = e.forward(xin, torch.cat([t, t], 0), **cond2)
            if erase_target is None:
                erase_target = torch.rand(noise.shape, device=e.device, dtype=torch.float32)
            tgt = torch.cat([noise.float(), erase_target.to(device=e.device, dtype=torch.float32)], 0)
            cot, _, sums = mse_bwd_seed(pred, tgt, scale)
            e.backward(cot, nsets=2)
            chw = pred[0].numel()
            self.last = dict(loss_x=sums[:B] / chw, loss_a=sums[B:] / chw)
        elif self.loss_fn in (NEG_GRAD, NAIVE):
            # a `loss` is returned (ddpm_deletion_loss.py:82-96): ONE backward, no gradient split
            # (delete_celeb.py:682-684); NegGrad ascends on the forget batch with -superfactor
            keep = self.loss_fn == NAIVE
            m = mixture_fwd(x0, a0, noise, t, torch.full((B,), 1.0 if keep else 0.0, device=e.device), self.ac,
                            self.gamma_tab, self.sigma_tab, 0.5)
            pred = e.forward(m.x_mix, t, **cond)
            cot, _, sums = mse_bwd_seed(pred, noise, scale if keep else -self.superfactor * scale)
            # the second gradient set stays zero (zero_grad above): g = g_x through the inf-guarded s = 0
            e.backward(cot, nsets=1)
            per = sums / pred[0].numel()
            # the returned 7-tuple (ddpm_deletion_loss.py:88,:96): naive = (loss_x, loss_x, None, ...), NegGrad = (-sf loss_a, None, loss_a, ...)
            self.last = dict(loss=per, loss_x=per) if keep else dict(loss=-self.superfactor * per, loss_a=per,
                                                                   superfactor=self.superfactor)
            if not keep and self.superfactor_decay is not None:
                self.superfactor *= self.superfactor_decay         # applied after the loss call (delete_celeb.py:658-662)
        elif self.loss_fn == SUBSCORE:
            # ddpm_deletion_loss.py:99-122: Bernoulli keep/forget rows against the plain noise target; the row
            # selection loss[mask] / (1 - lambd), loss[~mask] is a per-sample weight on the shared forward
            uu = u.to(device=e.device, dtype=torch.float32)
            m = mixture_fwd(x0, a0, noise, t, uu, self.ac, self.gamma_tab, self.sigma_tab, self.lambd)
            pred = e.forward(m.x_mix, t, **cond)
            c, _, sums = mse_bwd_seed(pred, noise, scale)
            keep = (uu > self.lambd).float()
            any_keep = (keep.sum() > 0).float()    # no keep rows: BOTH losses are zeroed (:113-116)
            wx = (keep / (1.0 - self.lambd)).view(B, 1, 1, 1)
            wa = ((1.0 - keep) * any_keep).view(B, 1, 1, 1)
            cot = e._buf("cot", (2 * B, *pred.shape[1:]))
            torch.mul(c, wx, out=cot[:B])
            torch.mul(c, wa, out=cot[B:])
            e.backward(cot, nsets=2)
            # loss[mask] / (1 - lambd) and loss[~mask] are ROW SELECTIONS (ddpm_deletion_loss.py:109-110): the logged
            # statistics run over the selected rows only; the zero-size guards (:113-120) log one zero
            self.last = dict(loss_x=sums / pred[0].numel() / (1.0 - self.lambd), loss_a=sums / pred[0].numel(),
                             rows_x=keep, rows_a=(1.0 - keep) * any_keep, subscore=True)
        else:
            raise ValueError(f"loss_fn {self.loss_fn!r} is not on the HIP fast path")
        self._micro += 1
        if self._micro == self.ga:
            self._micro = 0
            self._sync_and_update()

    def _partials(self, B, chw):
        return self.e._buf("loss_partials", (lib.query("siss_loss_partials_words", B, chw),), torch.float64)

    def _early_allreduce(self):
        if self._micro + 1 != self.ga:          # only the sync micro-step communicates
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the overlapped gradient exchange issues async collectives from inside the backward pass "
                               "and cannot be captured into a hipGraph: run N > 1 eagerly or set_overlap(False)")
        ps = self.e.ps
        # ONE grouped collective: the early-final tails of both gradient sets
        self._pending.append(allreduce_pieces([ps.grads[k, ps.split:] for k in range(ps.grads.shape[0])], self.pg, async_op=True))

    def _sync_and_update(self):
        g = self.e.ps.grads
        sharded = (self.dp_on and self.exchange == "sharded" and not self.overlap
                   and can_shard(g.shape[1], self.world))
        if not sharded:
            self._gather_optimizer_state()      # a replicated update after sharded ones: every rank needs the whole m / v
        if self.dp_on and not sharded:
            # the exchange of the step: sum of [g_x ; g_a] over ranks (RCCL over xGMI) -- one flat buffer; with
            # the overlap hook the early-final tail is already in flight and only the head remains
            if self.overlap and self._pending:
                split = self.e.ps.split                 # ... the second (and last) grouped collective of the step: the heads
                self._pending.append(allreduce_pieces([g[k, :split] for k in range(g.shape[0])], self.pg, async_op=True))
                for w in self._pending:
                    if w is not None:
                        w.wait()
                self._pending = []
            else:
                (EXCHANGES.get(self.exchange) or allreduce_flat_grads)(g, self.pg)     # 'sharded' that cannot shard: all-reduce
        single = self.loss_fn in (NEG_GRAD, NAIVE)
        okw = dict(scaling_norm=self.scaling_norm if not single else 1.0,
                   eta=self.eta if self.loss_fn == ERASEDIFF else None, inf_guard=self.inf_guard or single)
        if sharded:
            # reduce-scatter -> this rank's 1/N of the norm sums (3 doubles all-reduced) and of the recombine / clip / AdamW
            # pass -> all-gather of the updated f32 parameters (3 P (N-1)/N floats on the wire instead of 4 P (N-1)/N,
            # optimizer traffic / N); the bf16 operand shadow is recast from the gathered master
            from .dp import all_gather_params, reduce_scatter_param_shards
            gx_s, ga_s, lo, hi = reduce_scatter_param_shards(g, self.pg)
            self.opt.launch_sharded(gx_s, ga_s, lo, hi, self.pg, **okw)
            self._state_shard = (lo, hi)                        # m / v are current on [lo, hi) only
            all_gather_params(self.e.ps.flat, lo, hi, self.pg)
            self.e.refresh_weights(cast_shadow=True)
            return
        self.opt.launch(g, **okw)
        self.e.refresh_weights(lazy=True)               # (the dgrad weight copies: beside the next forward pass)

    def step(self, x0, a0, noise, t, u, conditioning=None, erase_target=None):
        """GA=1 convenience."""
        assert self.ga == 1
        self.micro_step(x0, a0, noise, t, u, conditioning, erase_target)

    # ------------------------------------------------------------------ logging (one small D2H)
    def stats(self):
        """The scalars the reference logs for the LAST micro-batch (delete_celeb.py:626-663: loss / loss_x / loss_a mean
        over all elements and max / min / unbiased std over the per-sample means; importance-weight mean / max / min /
        std; superfactor) plus the optimizer step's gradient scalars (:748) -- fetched with ONE device-to-host copy."""
        return self.stats_async().get()

    def stats_async(self):
        """The same without stalling the launch stream: the scalars are gathered on the device and copied to pinned host memory
        behind the step's kernels NOW; `.get()` waits for that copy and forms the dictionary.  A training loop calls `.get()`
        one step later (siss_amd/tasks.py), so the device always has the next step queued behind the one it is running."""
        last = dict(self.last or {})
        keys = [k for k in ("loss", "loss_x", "loss_a", "iw_x", "iw_a", "rows_x", "rows_a") if last.get(k) is not None]
        sizes = [self.opt.scalars.numel()] + [last[k].numel() for k in keys]
        dev = torch.cat([self.opt.scalars.float().flatten()] + [last[k].float().flatten() for k in keys])
        if dev.is_cuda:
            host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
            host.copy_(dev, non_blocking=True)                  # the one D2H of the step
            done = torch.cuda.Event()
            done.record()
        else:
            host, done = dev.clone(), None
        return _PendingStats(host, done, keys, sizes, last.get("subscore"), last.get("superfactor") if "superfactor" in last else None,
                             self.opt.stats_from)


class _PendingStats:
    """Handle of SISSStepper.stats_async()."""

    def __init__(self, host, done, keys, sizes, subscore, superfactor, opt_stats_from):
        self.host, self.done, self.keys, self.sizes = host, done, keys, sizes
        self.subscore, self.superfactor, self.opt_stats_from = subscore, superfactor, opt_stats_from

    def get(self):
        import math
        if self.done is not None:
            self.done.synchronize()
        host = self.host
        st = self.opt_stats_from(host[:self.sizes[0]])
        off = self.sizes[0]
        vals = {}
        for k, n in zip(self.keys, self.sizes[1:]):
            vals[k] = host[off:off + n].double()
            off += n

        def block(name, v):
            st[name + "/mean"] = float(v.mean())
            st[name + "/max"], st[name + "/min"] = float(v.max()), float(v.min())
            st[name + "/std"] = float(v.std()) if v.numel() > 1 else math.nan      # torch.std of one value: nan, as logged

        for k in ("loss", "loss_x", "loss_a"):
            if k in vals:
                v = vals[k]
                if self.subscore and k != "loss":
                    v = v[vals["rows_" + k[-1]] > 0]
                    if v.numel() == 0:
                        v = torch.zeros(1, dtype=torch.float64)                 # ddpm_deletion_loss.py:113-120
                block(k, v)
        for k in ("iw_x", "iw_a"):
            if k in vals:
                block("importance_weight_" + k[-1], vals[k])
        if self.superfactor is not None:
            st["superfactor"] = self.superfactor
        return st
