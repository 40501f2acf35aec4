"""A small Hydra-compatible config loader (hydra / omegaconf are not available offline).

Covers what the reference's entry point uses (main.py:15-35, config/*.yaml): ``defaults:`` lists
with ``_self_``, ``${a.b}`` / ``${a.b[0]}`` interpolation, attribute + item access, ``key=value``
command-line overrides, ``--config-name`` / ``--config-path``, and ``instantiate`` of ``_target_``
nodes with a remap table so that ``diffusers.UNet2DModel`` / ``torch.optim.AdamW`` /
``diffusers.DDPMScheduler`` resolve to this package's MI355X-native classes.  The reference's own
YAML files load unchanged.
"""
import importlib
import os
import re

import yaml


class Cfg(dict):
    """dict with attribute access and lazy ${...} interpolation against the root."""

    def __init__(self, data=None, root=None):
        super().__init__()
        object.__setattr__(self, "_root", root if root is not None else self)
        for k, v in (data or {}).items():
            super().__setitem__(k, _wrap(v, self._root))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v, self._root))

    def __getitem__(self, k):
        return _resolve(super().__getitem__(k), self._root)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def to_dict(self):
        return {k: _plain(self[k]) for k in self.keys()}


class CfgList(list):
    def __init__(self, data, root):
        super().__init__(_wrap(v, root) for v in data)
        self._root = root

    def __getitem__(self, i):
        v = super().__getitem__(i)
        return _resolve(v, self._root) if not isinstance(i, slice) else v

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


def _wrap(v, root):
    if isinstance(v, Cfg) or isinstance(v, CfgList):
        return v
    if isinstance(v, dict):
        return Cfg(v, root)
    if isinstance(v, (list, tuple)):
        return CfgList(v, root)
    if isinstance(v, str) and _SCI_FLOAT.match(v):
        return float(v)           # "1e-5": PyYAML (YAML 1.1) reads a string, OmegaConf (YAML 1.2 floats) a float
    return v


_SCI_FLOAT = re.compile(r"^[+-]?\d+(\.\d*)?[eE][+-]?\d+$")


def _plain(v):
    if isinstance(v, Cfg):
        return v.to_dict()
    if isinstance(v, list):
        return [_plain(x) for x in v]
    return v


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _select(root, path):
    cur = root
    for part in re.findall(r"[^.\[\]]+", path):
        if isinstance(cur, list):
            cur = cur[int(part)]
        else:
            cur = cur[part]
    return cur


def _resolve(v, root):
    if not isinstance(v, str) or "${" not in v:
        return v
    m = _INTERP.fullmatch(v)
    if m:                                   # whole-value interpolation keeps the type
        return _select(root, m.group(1))
    return _INTERP.sub(lambda mm: str(_select(root, mm.group(1))), v)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_yaml(config_path, name):
    fn = os.path.join(config_path, name if name.endswith(".yaml") else name + ".yaml")
    with open(fn) as f:
        return yaml.safe_load(f) or {}


def compose(config_name, config_path="config", overrides=()):
    """Hydra-style composition: ``defaults`` first (in order), ``_self_`` where listed (last if absent)."""
    raw = _load_yaml(config_path, config_name)
    defaults = raw.pop("defaults", [])
    merged = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(merged, raw)
            self_done = True
        elif isinstance(d, str):
            _merge(merged, compose(d, config_path).to_dict_raw())
        elif isinstance(d, dict):            # group/option form:  - optimizer: adamw
            for grp, opt in d.items():
                _merge(merged.setdefault(grp, {}), _load_yaml(os.path.join(config_path, grp), opt))
    if not self_done:
        _merge(merged, raw)
    for ov in overrides:
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        cur = merged
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = yaml.safe_load(val)
    cfg = Cfg(merged)
    object.__setattr__(cfg, "_raw", merged)
    return cfg


def _to_dict_raw(self):
    return getattr(self, "_raw", None) or {k: dict.__getitem__(self, k) for k in self.keys()}


Cfg.to_dict_raw = _to_dict_raw

# _target_ remap: the reference's third-party classes -> this package's MI355X-native ones
TARGET_REMAP = {
    "diffusers.UNet2DModel": "siss_amd.model.UNet2DModel",
    "diffusers.DDPMScheduler": "siss_amd.scheduler.DDPMScheduler",
    "torch.optim.AdamW": "siss_amd.optim.AdamWSpec",
    "torchvision.transforms.Compose": "siss_amd.data.Compose",
    "torchvision.transforms.ToTensor": "siss_amd.data.ToTensor",
    "torchvision.transforms.Normalize": "siss_amd.data.Normalize",
    "data.src.celeb_dataset.CelebAHQ": "siss_amd.data.CelebAHQ",
}


def get_object(path):
    path = TARGET_REMAP.get(path, path)
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node, *args, _recursive_=True, **kwargs):
    """hydra.utils.instantiate for ``_target_`` nodes (main.py:30-34 uses _recursive_=False)."""
    if node is None:
        return None
    target = node["_target_"]
    kw = {}
    for k in node.keys():
        if k.startswith("_"):
            continue
        v = node[k]
        if _recursive_:
            v = _inst_rec(v)
        kw[k] = v
    kw.update(kwargs)
    return get_object(target)(*args, **kw)


def _inst_rec(v):
    if isinstance(v, dict) and "_target_" in v:
        return instantiate(v)
    if isinstance(v, dict):
        return {k: _inst_rec(v[k]) for k in v.keys()}
    if isinstance(v, list):
        return [_inst_rec(x) for x in v]
    return v
