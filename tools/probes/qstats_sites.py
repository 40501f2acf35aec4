"""How many GroupNorm forward sites of the full-size CelebA-HQ network take their statistics from the producing conv."""
import sys
import torch
sys.path.insert(0, ".")
from siss_amd import lib
from siss_amd.config import UNet2DConfig
from siss_amd.unet import UNetEngine

lib.load()
eng = UNetEngine(UNet2DConfig.celebahq256(), "cuda:0")
eng.init_random(seed=0)
x = torch.randn(16, 3, 256, 256, device="cuda:0")
t = torch.full((16,), 999, dtype=torch.int64, device="cuda:0")
lib.dispatch_counts(reset=True)
lib.PROF = []
eng.forward(x, t)
torch.cuda.synchronize()
cnt = lib.dispatch_counts(reset=True)
gn = [r for r in lib.PROF if r[0] == "siss_groupnorm_fwd"]
big = [r for r in gn if dict(zip(r[4][::2], r[4][1::2]))["H"] > 32]
print("GroupNorm forward launches", len(gn), "of them at more than 32 x 32 pixels", len(big),
      "statistics from the producer", cnt["gn_qstats"], "slab kernels", cnt["gn_slab"])
