"""Steady-state forward pass of the fused HoVer-Net (batch 32 x 256^2) for a rocprofv3 kernel trace."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tiatoolbox_amd.models.architecture import get_pretrained_model  # noqa: E402
from tiatoolbox_amd.models.architecture.hovernet_fused import FusedHoVerNet  # noqa: E402
from tiatoolbox_amd.utils import synth  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
plain = len(sys.argv) > 2 and sys.argv[2] == "plain"
model, _ = get_pretrained_model("hovernet_fast-pannuke")
model = model.eval().cuda()
net = model.to(memory_format=torch.channels_last) if plain else FusedHoVerNet(model).cuda()
x = torch.from_numpy(synth.g_he(batch, 256, 256, seed=5)).cuda().float().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
with torch.inference_mode():
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        net(x)
    torch.cuda.synchronize()
print(f"{'plain' if plain else 'fused'} forward: {(time.perf_counter() - t0) / 4 / batch * 1e3:.3f} ms per tile (batch {batch})")
