"""Task classes behind the reference's entry points (main.py:9-35): ``Task.run()``,
``delete_celeb.DeleteCeleb(cfg)``, ``delete_tshirt.DeleteTShirt(cfg)``, ``delete_sd.DeleteSD(cfg)``.

The loop is the hot path of delete_celeb.py:557-773 / delete_tshirt.py:501-717 driven through
``SISSStepper`` (fused kernels, one dual-cotangent backward, one collective per optimizer step).
Evaluation / sampling / wandb (log_metrics, delete_celeb.py:484-545) are outside this path; the
per-step scalars the reference logs are returned by ``stepper.stats()`` and written as JSON lines.
"""
import json
import os
import time
from abc import ABC, abstractmethod

import torch

from . import hydra_lite
from .config import UNet2DConditionConfig, UNet2DConfig
from .data import InfiniteSampler, Prefetcher, RepeatedSampler, SyntheticImages, batches
from .scheduler import DDPMScheduler
from .step import SISSStepper


class Task(ABC):
    @abstractmethod
    def run(self):
        pass


def _dist():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    pg = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        pg = dist.group.WORLD
    return world, rank, local, pg


class _DeleteBase(Task):
    timestep_low = 999          # delete_celeb.py:593 hard-codes randint(999, 1000)
    inf_guard = False
    default_unet = staticmethod(UNet2DConfig.celebahq256)

    def __init__(self, cfg):
        self.cfg = cfg

    # -- pieces a subclass may override --------------------------------------------------------
    def compute_dtype(self):
        """`mixed_precision: null` -- the reference's shipped default: fp32 everywhere (delete_celeb.yaml:103) -- runs the f32 engine
        (every tensor and product f32: the parity mode, 1/16 of the bf16 MFMA rate at best); `bf16` the bf16 MFMA path."""
        mp = self.cfg.get("mixed_precision")
        if mp == "bf16":
            return torch.bfloat16
        if mp not in (None, "no", "null", "None"):
            # the reference's YAML lists fp16 / fp8 as choices of accelerate; there is no fp16 path here (and no loss scaler in the
            # reference's loop either): refuse instead of silently running something else
            raise NotImplementedError(f"mixed_precision={mp!r}: this build computes in 'bf16' or in f32 (null / 'no'); "
                                      "fp16 / fp8 are not provided")
        print("[siss_amd] mixed_precision is null / 'no': true f32 compute for the UNet, the loss and the optimizer (the reference's "
              "default; a parity mode here, far slower than mixed_precision=bf16, which is what the benchmark runs; the SD front end -- "
              "VAE / text encoder -- stays bf16)")
        return torch.float32

    def load_unet(self, device):
        from .model import UNet2DModel
        cfg = self.cfg
        path = cfg.get("checkpoint_path")
        dt = self.compute_dtype()
        if path and os.path.isdir(str(path)):
            sub = (cfg.get("subfolders") or {}).get("unet")
            return UNet2DModel.from_pretrained(path, subfolder=sub, device=device, compute_dtype=dt)
        # The reference hard-fails here (DDPMPipeline.from_pretrained, delete_celeb.py:181).  Random-init weights of the
        # same architecture are for benchmarks / smoke runs only and must be asked for: allow_random_init=true.
        if not cfg.get("allow_random_init"):
            raise FileNotFoundError(
                f"checkpoint_path {path!r} is not a directory on disk (no network: hub ids cannot be fetched); "
                "pass allow_random_init=true to train random-init weights of the configured architecture instead")
        ucfg = {k: v for k, v in (cfg.get("unet") or {}).items() if not k.startswith("_")}
        m = UNet2DModel(UNet2DConfig.from_dict(ucfg) if ucfg else self.default_unet(), device=device, compute_dtype=dt)
        m.engine.init_random(seed=int(cfg.random_seed))
        print(f"[siss_amd] allow_random_init: checkpoint {path!r} not on disk, RANDOM-INIT weights of the same architecture")
        return m

    def load_scheduler(self):
        sc = self.cfg.get("scheduler") or {}
        kw = {k: sc[k] for k in ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type")
              if k in sc}
        return DDPMScheduler(**kw)

    def datasets(self, shape):
        cfg = self.cfg
        try:
            if cfg.get("dataset_all") is None or cfg.get("dataset_deletion") is None:
                raise FileNotFoundError("no dataset_all / dataset_deletion configured")
            transform = hydra_lite.instantiate(cfg.transform)
            ds_all = hydra_lite.instantiate(cfg.dataset_all, transform=transform)
            ds_del = hydra_lite.instantiate(cfg.dataset_deletion, transform=transform)
            return ds_all, ds_del
        except FileNotFoundError as e:
            # Only a MISSING dataset can be replaced, and only when asked for (allow_synthetic=true: benchmarks / smoke
            # runs; the datasets are not in the repo, reference .gitignore:7).  Every other error -- a typo in a target,
            # a bad transform, a decode failure -- surfaces, as it does in the reference.
            if not cfg.get("allow_synthetic"):
                raise
            print(f"[siss_amd] allow_synthetic: dataset unavailable ({e}); using SYNTHETIC images")
            return SyntheticImages(4096, shape, seed=1), SyntheticImages(1, shape, seed=2)

    def check_supported(self):
        """Refuse, loudly, the configured features of the reference loop this loop does not implement -- rather than
        run something else than what the config says."""
        cfg = self.cfg
        ema = cfg.get("ema") or {}
        if ema.get("use_ema") or cfg.get("use_ema"):
            raise NotImplementedError("ema.use_ema=true: EMA of the unlearned weights is not implemented "
                                      "(delete_celeb.py:235-251; every shipped delete config has it off)")
        for key in ("checkpointing_steps", "resume_from_checkpoint"):
            if cfg.get(key) not in (None, "null", False, 0):
                raise NotImplementedError(f"{key}={cfg.get(key)!r}: accelerate-state checkpoints are not implemented "
                                          "(delete_celeb.py:786-825; null in every shipped delete config); the final model "
                                          "is written in the diffusers layout")
        if cfg.get("mixed_precision") not in (None, "null", "no", "bf16"):
            raise NotImplementedError(f"mixed_precision={cfg.get('mixed_precision')!r}: only null / bf16 (no loss scaler)")

    def optimizer_args(self):
        """(lr, betas, eps, weight_decay) of the AdamW the reference instantiates from cfg.optimizer
        (delete_celeb.py:232)."""
        opt = hydra_lite.instantiate(self.cfg.optimizer)
        return opt.lr, opt.betas, opt.eps, opt.weight_decay

    def conditioning(self, B, device):
        """The dict the reference splats into the UNet call (delete_celeb.py:622: {})."""
        return None

    def lr_schedule_args(self, world):
        """(num_warmup_steps, num_training_steps) as the reference hands them to get_scheduler
        (delete_celeb.py:296-301: cfg.warmup_steps, cfg.training_steps)."""
        cfg = self.cfg
        return int(cfg.get("warmup_steps") or 0), int(cfg.get("training_steps") or 0)

    def sample_noise(self, shape, device, generator):
        """noise = torch.randn_like(images) (delete_celeb.py:583); the SAME noise for the keep and the forget batch."""
        return torch.randn(shape, device=device, generator=generator)

    def seed(self):
        return int(self.cfg.random_seed)

    def prepare_batch(self, x, generator):
        """Images -> what the loss is taken on (identity for the pixel-space tasks; VAE latents for SD)."""
        return x

    def evaluate(self, unet, sched, forget_image, step, device):
        """The image part of the reference's log_metrics (delete_celeb.py:376-436, :484-545): `eval_batch_size`
        samples from the current model (DDPMPipeline, `pipeline.num_inference_steps`) and the forget image noised to
        `metrics.denoising_injections.timestep` and denoised back, written as PNG grids instead of wandb images.
        Runs every `eval_every` optimizer steps -- OPT-IN (null by default): the reference's `sampling_steps: 1`
        spends ~300 UNet forwards per optimizer step on it."""
        import numpy as np
        from PIL import Image
        from .sampler import Evaluator
        cfg = self.cfg
        ev = Evaluator(cfg)
        ev.load_model(unet, sched)
        n = int(cfg.get("eval_batch_size") or 4)
        steps = int(((cfg.get("pipeline") or {}).get("num_inference_steps")) or 50)
        imgs = ev.sample_images(n, num_inference_steps=steps, set_generator=True)            # [n, H, W, C] in [0, 1]
        inj = ((cfg.get("metrics") or {}).get("denoising_injections")) or {}
        t_inj = int(inj.get("timestep", 250))
        g = torch.Generator(device=device).manual_seed(self.seed())
        clean = forget_image.to(device).float()
        noise = torch.randn((n, *clean.shape), generator=g, device=device)
        noisy = sched.add_noise(clean.expand(n, *clean.shape), noise, torch.full((n,), t_inj, device=device))
        den = ev.denoise_images(noisy, t_inj).cpu().numpy()

        def grid(a):
            a = (np.clip(a, 0, 1) * 255).astype(np.uint8)
            row = np.concatenate(list(a), axis=1)
            return Image.fromarray(row[..., 0] if row.shape[-1] == 1 else row)
        grid(imgs).save(os.path.join(cfg.output_dir, f"samples_step{step}.png"))
        grid(den).save(os.path.join(cfg.output_dir, f"denoised_forget_t{t_inj}_step{step}.png"))

    # -- the loop ------------------------------------------------------------------------------
    def run(self):
        cfg = self.cfg
        world, rank, local, pg = _dist()
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        seed = self.seed()
        torch.manual_seed(seed + rank)
        self.check_supported()
        unet = self.load_unet(device)
        eng = unet.engine
        sched = self.load_scheduler()
        lr, betas, eps, wd = self.optimizer_args()
        from .scheduler import lr_multiplier
        lr_name = str(cfg.get("lr_scheduler") or "constant")
        # get_scheduler(cfg.lr_scheduler, num_warmup_steps=..., num_training_steps=...): delete_celeb.py:296-301 passes
        # cfg.warmup_steps / cfg.training_steps, delete_sd.py:714-719 cfg.lr_warmup_steps * num_processes / cfg.training_steps
        warm, total = self.lr_schedule_args(world)
        lr_multiplier(lr_name, 0, warm, total)                  # raises now for a schedule that is not implemented
        d = cfg.deletion
        B, ga = int(cfg.train_batch_size), int(cfg.gradient_accumulation_steps)
        stepper = SISSStepper(
            eng, sched.alphas_cumprod, lr=lr, betas=betas, eps=eps, weight_decay=wd,
            scaling_norm=float(d.scaling_norm) if d.loss_fn != "erasediff" else None,
            eta=float(d.eta) if d.loss_fn == "erasediff" else None,
            lambd=float((d.loss_params or {}).get("lambd", 0.5)), train_batch_size=B, grad_accum=ga,
            loss_fn=d.loss_fn, inf_guard=self.inf_guard, process_group=pg,
            mixed_precision=cfg.get("mixed_precision"),
            superfactor=float((d.loss_params or {}).get("superfactor", 1.0)),
            superfactor_decay=d.get("superfactor_decay"))
        shape = (unet.config.in_channels, unet.config.sample_size, unet.config.sample_size)
        ds_all, ds_del = self.datasets(shape)
        # keep shard: decoded / pinned / copied in the background, `depth` batches ahead (the forget set is one or a
        # few images repeated: a plain iterator is enough)
        it_all = Prefetcher(ds_all, InfiniteSampler(ds_all, rank=rank, num_replicas=world), B, device=device,
                            workers=int(cfg.get("dataloader_num_workers") or 4))
        it_del = batches(ds_del, self.deletion_sampler(ds_del, B), B)
        n_steps = int(cfg.training_steps) * max(1, len(d.get("img_name") or [1]))
        cond = self.conditioning(B, device)
        os.makedirs(cfg.output_dir, exist_ok=True)
        log = open(os.path.join(cfg.output_dir, f"train_log_rank{rank}.jsonl"), "a")
        g = torch.Generator(device=device).manual_seed(seed + rank)
        T = sched.config.num_train_timesteps
        t0 = time.perf_counter()
        eval_every = int(cfg.get("eval_every") or 0)
        pending = None

        def write_log(handle, meta):
            st = handle.get()
            st["global_step"], st["lr"], st["elapsed_s"] = meta          # (elapsed: stamped when the step was queued)
            log.write(json.dumps(st) + "\n")
            log.flush()
            if rank == 0:
                print(f"step {meta[0]}/{n_steps}  |g_x| {st['norm_loss_x']:.4g}  |g_a| {st['norm_loss_a']:.4g}  "
                      f"s {st['scaling_factor']:.4g}")

        try:
            for step in range(n_steps):
                # lr_scheduler.step() after every optimizer step (delete_celeb.py:770); accelerate's wrapper advances the
                # schedule once per PROCESS per optimizer step (AcceleratedScheduler, split_batches=False)
                stepper.opt.lr = lr * lr_multiplier(lr_name, step * world, warm, total)
                for _ in range(ga):
                    x0 = self.prepare_batch(next(it_all).to(device, non_blocking=True), g)
                    a0 = self.prepare_batch(next(it_del).to(device, non_blocking=True), g)
                    noise = self.sample_noise(x0.shape, device, g)                      # SAME noise for both batches
                    t = torch.randint(self.timestep_low, T, (B,), device=device, generator=g)
                    u = torch.rand(B, device=device, generator=g)
                    stepper.micro_step(x0, a0, noise, t, u, cond)
                # the step's scalars travel to the host behind its kernels; they are read (and logged) once the NEXT step is queued,
                # so the device never waits for the host between steps (same lines, written one step later)
                handle, meta = stepper.stats_async(), (step + 1, stepper.opt.lr, time.perf_counter() - t0)
                if pending is not None:
                    write_log(*pending)
                pending = (handle, meta)
                if eval_every and (step + 1) % eval_every == 0 and not isinstance(self, DeleteSD):
                    write_log(*pending)                 # every rank flushes at an evaluation step (the logs stay in step)
                    pending = None
                    if rank == 0:
                        self.evaluate(unet, sched, ds_del[0], step + 1, device)
        finally:
            # the last COMPLETED optimizer step reaches the log also when the next one raises or the job is interrupted
            if pending is not None:
                try:
                    write_log(*pending)
                except Exception:
                    pass
        it_all.close()
        if rank == 0 and cfg.get("save_final", True):
            unet.save_pretrained(os.path.join(cfg.output_dir, "unet"))
        return stepper

    def deletion_sampler(self, ds, B):
        return InfiniteSampler(ds, shuffle=False) if len(ds) >= B else RepeatedSampler(ds, 1 << 30)


class DeleteCeleb(_DeleteBase):
    """config/delete_celeb.yaml -- CelebA-HQ 256, forget set = listed jpgs repeated (RepeatedSampler)."""


class DeleteTShirt(_DeleteBase):
    """config/delete_tshirt.yaml -- MNIST + T-shirt: t ~ U{0..999} (delete_tshirt.py:535-540) and the
    inf guard on the scaling factor (:688-690)."""
    timestep_low = 0
    inf_guard = True
    default_unet = staticmethod(UNet2DConfig.mnist_tshirt)


class DeleteSD(_DeleteBase):
    """config/delete_sd.yaml -- Stable Diffusion: the same loop on VAE latents with text conditioning
    (delete_sd.py:864-1127).  On the path: the SISS step through ``UNet2DConditionModel``.  Not on it (SURVEY.md §8f
    rank 4, not built): the frozen VAE encoder and CLIP text encoder of delete_sd.py:879-888,:941-944 -- latents and the
    prompt embedding are read from ``.pt`` files when configured (``latents_all`` / ``latents_deletion`` [N,4,h,w],
    already multiplied by vae.config.scaling_factor; ``validation_prompts[0]`` ending in .pt [77,768] or [1,77,768],
    the reference's own ``using_augmented_prompt`` branch, delete_sd.py:938) and synthetic otherwise."""
    default_unet = staticmethod(UNet2DConditionConfig.sd15)
    VAE_SCALE = 0.18215            # vae.config.scaling_factor (delete_sd.py:883,888)
    vae = None
    text_encoder = None

    def load_front_end(self, device):
        """Frozen VAE encoder + CLIP text encoder from the checkpoint directory, when it is on disk
        (delete_sd.py:464-474): images are then encoded per micro-batch and the prompt once."""
        path = str(self.cfg.get("pretrained_model_name_or_path") or "")
        if os.path.isdir(os.path.join(path, "vae")):
            from .vae import VAEEncoder
            self.vae = VAEEncoder.from_pretrained(path, "vae", device)
        if os.path.isdir(os.path.join(path, "text_encoder")):
            from .text_encoder import CLIPTextEncoder
            self.text_encoder = CLIPTextEncoder.from_pretrained(path, "text_encoder", device)

    def prepare_batch(self, x, generator):
        if self.vae is not None and x.shape[1] == self.vae.cfg.in_channels and x.shape[1] != 4:
            return self.vae.encode(x, generator=generator)      # latent_dist.sample() * scaling_factor (:879-888)
        return x

    def seed(self):
        return int(self.cfg.get("seed", 42))                   # config/delete_sd.yaml:80

    def load_unet(self, device):
        from .model import UNet2DConditionModel
        cfg = self.cfg
        self.load_front_end(device)
        path = cfg.get("pretrained_model_name_or_path")
        dt = self.compute_dtype()                                # (the frozen VAE / text encoder front end stays on the bf16 path)
        if path and os.path.isdir(os.path.join(str(path), "unet")):
            return UNet2DConditionModel.from_pretrained(path, subfolder="unet", device=device, compute_dtype=dt)   # delete_sd.py:458-462
        if not cfg.get("allow_random_init"):                     # the reference hard-fails (from_pretrained, delete_sd.py:458-462)
            raise FileNotFoundError(
                f"pretrained_model_name_or_path {path!r} has no unet/ directory on disk (no network: hub ids cannot be "
                "fetched); pass allow_random_init=true to train random-init weights of the configured architecture instead")
        ucfg = {k: v for k, v in (cfg.get("unet") or {}).items() if not k.startswith("_")}
        m = UNet2DConditionModel(UNet2DConditionConfig.from_dict(ucfg) if ucfg else self.default_unet(), device=device, compute_dtype=dt)
        m.engine.init_random(seed=self.seed())
        print(f"[siss_amd] allow_random_init: {path!r}/unet not on disk, RANDOM-INIT weights of the same architecture")
        return m

    def load_scheduler(self):
        # the SD v1 scheduler_config.json: scaled-linear betas 0.00085 -> 0.012 (delete_sd.py:412)
        return DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")

    def check_supported(self):
        """delete_sd.py reads more knobs than the pixel-space loops; the ones this loop does not implement are refused
        instead of silently training something else (all at their shipped defaults in config/delete_sd.yaml:99-102,125)."""
        super().check_supported()
        cfg = self.cfg
        if cfg.get("input_perturbation"):
            raise NotImplementedError(f"input_perturbation={cfg.get('input_perturbation')!r}: perturbed noising with an unperturbed "
                                      "target (delete_sd.py:899-903,:920-926) is not implemented; null in config/delete_sd.yaml")
        if cfg.get("snr_gamma") is not None:
            raise NotImplementedError(f"snr_gamma={cfg.get('snr_gamma')!r}: the reference's only live loss branch is "
                                      "`if self.cfg.snr_gamma is None` (delete_sd.py:962); min-SNR weighting is not implemented")
        if cfg.get("prediction_type") not in (None, "null", "epsilon"):
            raise NotImplementedError(f"prediction_type={cfg.get('prediction_type')!r}: only the epsilon objective is on this "
                                      "path (the v-prediction target is commented out in the reference too: delete_sd.py:953-958)")
        if cfg.get("use_8bit_adam"):
            raise NotImplementedError("use_8bit_adam=true: bitsandbytes' 8-bit AdamW (delete_sd.py:555-565) is not implemented; "
                                      "the fused step keeps f32 moments")

    def lr_schedule_args(self, world):
        cfg = self.cfg                                          # delete_sd.py:714-719
        return int(cfg.get("lr_warmup_steps") or 0) * world, int(cfg.get("training_steps") or 0)

    def sample_noise(self, shape, device, generator):
        noise = torch.randn(shape, device=device, generator=generator)
        off = float(self.cfg.get("noise_offset") or 0.0)
        if off:                                                 # delete_sd.py:893-898 (offset noise, one draw per (sample, channel))
            noise = noise + off * torch.randn((shape[0], shape[1], 1, 1), device=device, generator=generator)
        return noise

    def optimizer_args(self):
        c = self.cfg                                            # delete_sd.py:546-552 (scale_lr), :567-573
        lr = float(c.learning_rate)
        if c.get("scale_lr"):
            world = int(os.environ.get("WORLD_SIZE", "1"))
            lr *= int(c.gradient_accumulation_steps) * int(c.train_batch_size) * world
        return lr, (float(c.adam_beta1), float(c.adam_beta2)), float(c.adam_epsilon), float(c.adam_weight_decay)

    def datasets(self, shape):
        cfg = self.cfg
        la, ld = cfg.get("latents_all"), cfg.get("latents_deletion")
        ia, idl = cfg.get("images_all"), cfg.get("images_deletion")
        from .data import TensorImages
        if self.vae is not None and ia and idl and os.path.exists(str(ia)) and os.path.exists(str(idl)):
            return TensorImages(torch.load(ia)), TensorImages(torch.load(idl))     # [N,3,H,W] in [-1,1]: encoded per batch
        if la and ld and os.path.exists(str(la)) and os.path.exists(str(ld)):
            return TensorImages(torch.load(la)), TensorImages(torch.load(ld))
        if not cfg.get("allow_synthetic"):
            raise FileNotFoundError("no images_all / images_deletion (with a vae/ on disk) or latents_all / latents_deletion "
                                    "tensor files configured; pass allow_synthetic=true for synthetic latents")
        print("[siss_amd] allow_synthetic: no image / latent files configured, SYNTHETIC latents")
        return (SyntheticImages(4096, shape, seed=1, scale=self.VAE_SCALE, normal=True),
                SyntheticImages(1, shape, seed=2, scale=self.VAE_SCALE, normal=True))

    def conditioning(self, B, device):
        cfg = self.cfg
        vp = cfg.get("validation_prompts")
        X = int((cfg.get("unet") or {}).get("cross_attention_dim", 768))
        if vp and str(vp[0]).endswith(".pt") and os.path.exists(str(vp[0])):
            e = torch.load(str(vp[0])).to(device)
            if not e.is_floating_point():                       # token ids [77] / [1,77]: run the text encoder
                assert self.text_encoder is not None, "token ids given but no text_encoder/ in the checkpoint directory"
                e = self.text_encoder(e.reshape(1, -1))[0]
            e = e.float().reshape(-1, e.shape[-2], e.shape[-1])[:1]
        elif vp and self.text_encoder is not None:
            from transformers import CLIPTokenizer               # delete_sd.py:395-410 tokenize_captions
            tok = CLIPTokenizer.from_pretrained(str(cfg.pretrained_model_name_or_path), subfolder="tokenizer")
            ids = tok([str(vp[0])], max_length=tok.model_max_length, padding="max_length", truncation=True,
                      return_tensors="pt").input_ids
            e = self.text_encoder(ids)[0].float()
        elif cfg.get("allow_synthetic"):
            e = torch.randn(1, 77, X, generator=torch.Generator().manual_seed(self.seed())).to(device)
        else:
            raise FileNotFoundError("validation_prompts[0] is neither a .pt embedding / token file on disk nor a prompt with a "
                                    "text_encoder/ + tokenizer/ on disk; pass allow_synthetic=true for a synthetic embedding")
        return {"encoder_hidden_states": e.repeat(B, 1, 1)}     # one prompt for the whole batch (delete_sd.py:941-944)
