"""Input side of the hot path: transforms, datasets, samplers.

Restates data/utils/infinite_sampler.py:4-35 and data/utils/repeat_sampler.py:4-21 with the
rank / num_replicas arguments actually wired (the reference never passes them, so all of its
ranks read identical batches -- SURVEY.md §5), plus a synthetic source for benchmarking.
"""
import os

import numpy as np
import torch


class ToTensor:
    def __call__(self, img):
        a = np.asarray(img, dtype=np.float32) / 255.0
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)

    def __call__(self, x):
        m = torch.tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
        s = torch.tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
        return (x - m) / s


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class CelebAHQ(torch.utils.data.Dataset):
    """Directory of jpgs with filter in {all, deletion, nondeletion} (data/src/celeb_dataset.py:5-36)."""

    def __init__(self, filter, data_path, remove_img_names, transform=None):
        from PIL import Image
        self._open = Image.open
        names = sorted(f for f in os.listdir(data_path) if f.lower().endswith((".jpg", ".png")))
        remove = set(remove_img_names)
        if filter == "deletion":
            names = [n for n in names if n in remove]
        elif filter == "nondeletion":
            names = [n for n in names if n not in remove]
        elif filter != "all":
            raise ValueError(filter)
        self.paths = [os.path.join(data_path, n) for n in names]
        self.transform = transform

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        img = self._open(self.paths[i]).convert("RGB")
        return self.transform(img) if self.transform else img


class SyntheticImages(torch.utils.data.Dataset):
    """x ~ U[-1,1] images of a fixed shape (ToTensor+Normalize(0.5,0.5) range), or scale * N(0,1) "latents"
    (normal=True: the SD task's stand-in for VAE latents); deterministic per index."""

    def __init__(self, n, shape, seed=0, scale=1.0, normal=False):
        self.n, self.shape, self.seed, self.scale, self.normal = n, tuple(shape), seed, float(scale), normal

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + int(i))
        if self.normal:
            return self.scale * torch.randn(self.shape, generator=g)
        return self.scale * (torch.rand(self.shape, generator=g) * 2 - 1)


class TensorImages(torch.utils.data.Dataset):
    """A stack [N, C, H, W] already in memory (pre-encoded SD latents)."""

    def __init__(self, t):
        assert t.dim() == 4
        self.t = t.float()

    def __len__(self):
        return self.t.shape[0]

    def __getitem__(self, i):
        return self.t[i]


class InfiniteSampler(torch.utils.data.Sampler):
    """Infinite windowed shuffle; rank r of R takes every R-th index."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield int(order[i])
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1


class RepeatedSampler(torch.utils.data.Sampler):
    """Each index num_repeats times, in order (the celeb forget set is one image repeated)."""

    def __init__(self, data_source, num_repeats):
        self.data_source, self.num_repeats = data_source, num_repeats

    def __len__(self):
        return len(self.data_source) * self.num_repeats

    def __iter__(self):
        return iter(torch.arange(len(self.data_source)).repeat_interleave(self.num_repeats).tolist())


def batches(dataset, sampler, batch_size):
    """Endless batch iterator (stack of dataset[i])."""
    buf = []
    while True:
        for i in sampler:
            buf.append(dataset[i])
            if len(buf) == batch_size:
                yield torch.stack(buf)
                buf = []


class Prefetcher:
    """Background batch pipeline for real-image runs: samples are decoded / transformed by a
    small thread pool (PIL decoding releases the GIL), stacked into PINNED host buffers and copied to the device on a
    dedicated copy stream, `depth` batches ahead of the consumer -- at 66 ms per step a synchronous loop would spend
    more time decoding 32 JPEGs than the GPU spends on the step.  The batch ORDER is exactly that of
    ``batches(dataset, sampler, batch_size)`` (one producer walks the sampler).  Iterating yields device tensors;
    on a CPU `device` it degrades to plain background loading."""

    def __init__(self, dataset, sampler, batch_size, device="cuda", depth=3, workers=4):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.dataset, self.sampler, self.batch_size = dataset, sampler, int(batch_size)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.q = queue.Queue(maxsize=max(1, depth))
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._stop = threading.Event()
        self._err = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            idx = []
            for i in self.sampler:
                if self._stop.is_set():
                    return
                idx.append(i)
                if len(idx) < self.batch_size:
                    continue
                items = list(self.pool.map(self.dataset.__getitem__, idx))
                idx = []
                host = torch.stack(items)
                if self.cuda:
                    host = host.pin_memory()
                    with torch.cuda.stream(self.copy_stream):
                        dev = host.to(self.device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.copy_stream)
                    item = (dev, ev, host)              # keep the pinned source alive until the copy is consumed
                else:
                    item = (host, None, None)
                while not self._stop.is_set():
                    try:
                        self.q.put(item, timeout=0.1)
                        break
                    except Exception:
                        continue
        except BaseException as e:                      # surfaced to the consumer, never swallowed
            self._err = e
            self.q.put(None)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            raise RuntimeError("data prefetch thread failed") from self._err
        dev, ev, _ = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            # The batch was allocated on the COPY stream's pool but is consumed on the compute stream -- by kernels that
            # take raw pointers (ctypes), which the caching allocator cannot see.  Without this the block returns to the
            # copy-stream pool the moment the caller drops the tensor and the producer thread's next host.to() may
            # overwrite it while the step's kernels (the loss seed re-reads x0 after the whole UNet forward) are still
            # queued.  record_stream defers the reuse until the compute stream has passed the point of the free.
            dev.record_stream(cur)
        return dev

    def close(self):
        self._stop.set()
        self.pool.shutdown(wait=False, cancel_futures=True)
