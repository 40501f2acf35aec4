#!/usr/bin/env python
"""One-command check of a REAL diffusers checkpoint on this build (needs a GPU):

    python tools/check_checkpoint.py /path/to/google-ddpm-celebahq-256            # UNet2DModel (config.json + safetensors)
    python tools/check_checkpoint.py /path/to/stable-diffusion-v1-5 --subfolder unet

Loads the checkpoint into the HIP engine (siss_amd.model.*.from_pretrained: strict key / shape match against the architecture
the config.json describes) AND into the fp32 torch restatement of the same network (oracle/unet.py, oracle/unet_cond.py), then
compares one forward and one dual-cotangent backward at B = 2 on the same inputs: prediction within 3e-2 of scale, every
tensor's gradient cosine >= 0.99 (the bar of tests/test_hip_large_kernels.py).  Every number in DESIGN.md is on random-init
weights of the exact architectures -- the build container has no network; this is the check for whoever has the files.
The oracle is test infrastructure: this tool is a checker, not a product path."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--subfolder", default=None)
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    from safetensors.torch import load_file
    from parity_util import assert_grads_match
    d = os.path.join(a.path, a.subfolder) if a.subfolder else a.path
    if not os.path.exists(os.path.join(d, "config.json")) and os.path.isdir(os.path.join(a.path, "unet")):
        d = os.path.join(a.path, "unet")
    cfg_json = json.load(open(os.path.join(d, "config.json")))
    cond = cfg_json.get("_class_name") == "UNet2DConditionModel" or "cross_attention_dim" in cfg_json
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    if cond:
        from siss_amd.model import UNet2DConditionModel as M
        from oracle.unet_cond import OracleUNet2DCondition as O, UNetCondConfig as OC
    else:
        from siss_amd.model import UNet2DModel as M
        from oracle.unet import OracleUNet2D as O, UNetConfig as OC
    hip = M.from_pretrained(d, device=dev)                          # raises on any missing / unexpected key or shape
    eng = hip.engine
    sd = load_file(os.path.join(d, "diffusion_pytorch_model.safetensors"))
    n_par = sum(v.numel() for v in sd.values())
    print(f"{'UNet2DConditionModel' if cond else 'UNet2DModel'}: {len(sd)} tensors, {n_par:,} parameters loaded into the HIP engine")
    fields = {f for f in OC.__dataclass_fields__}
    hcfg = vars(hip.config)
    net = O(OC(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in hcfg.items() if k in fields}))
    net.load_state_dict({k: v.float() for k, v in sd.items()})
    net = net.to(dev).float()
    B, c, hw = a.batch, hip.config.in_channels, hip.config.sample_size
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.randn(B, c, hw, hw, generator=g, device=dev) * (0.18215 if cond else 1.0)).to(torch.bfloat16)
    t = torch.tensor(([999, 250] * B)[:B], device=dev)
    cx, ca = (torch.randn(B, hip.config.out_channels, hw, hw, generator=g, device=dev) * 1e-3 for _ in range(2))
    kw = {}
    if cond:
        kw["encoder_hidden_states"] = torch.randn(B, 77, hip.config.cross_attention_dim, generator=g, device=dev).to(torch.bfloat16)
    pred = eng.forward(x, t, **kw).clone()
    eng.zero_grad()
    eng.backward(torch.cat([cx, ca]).contiguous(), nsets=2)
    torch.cuda.synchronize()
    ref = net(x.float(), t, *([kw["encoder_hidden_states"].float()] if cond else []))[0]
    err, scale = (pred - ref.detach()).abs().max().item(), ref.detach().abs().max().item()
    print(f"forward: max |pred - ref| = {err:.4g} at scale {scale:.4g} (rel {err / scale:.3g}; bar 3e-2)")
    assert err <= 3e-2 * scale
    names = [n for n, _ in net.named_parameters()]
    params = [p for _, p in net.named_parameters()]
    grads = [torch.autograd.grad(ref, params, c_, retain_graph=(i == 0)) for i, c_ in enumerate((cx, ca))]
    # (a mathematically zero gradient -- attention key biases -- must be negligible: 1e-5 of the total; bf16 noise on small nets is ~2e-6)
    worst = assert_grads_match(eng, names, grads, dev, zero_tol=1e-5)
    print(f"dual backward: worst per-tensor gradient cosine {worst[0]:.5f} at {worst[1]} (bar 0.99) -- checkpoint OK")


if __name__ == "__main__":
    main()
