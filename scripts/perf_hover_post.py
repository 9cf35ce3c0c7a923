"""HoVer-Net post-processing on the bench's synthetic head maps (256 tiles of 164^2, ~60 nuclei each) -- for a rocprofv3 kernel
trace (launch list and time split of `_proc_np_hv` + instance statistics + contours) and HIP-event timing of each stage."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from tiatoolbox_amd.utils import synth
from tiatoolbox_amd.models.architecture import _hover_device as hd
from tiatoolbox_amd.models.architecture.hovernet import HoVerNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
npm, hv, tp = synth.hover_head_maps(8, 164, 164, seed=1, n_blobs=60)
dev = torch.device("cuda")
npm_d = torch.from_numpy(npm).to(dev).repeat(n // 8, 1, 1, 1)
hv_d = torch.from_numpy(hv).to(dev).repeat(n // 8, 1, 1, 1)
tp_d = torch.from_numpy(tp).to(dev).repeat(n // 8, 1, 1, 1)

def ev(fn, reps=reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

model = HoVerNet(num_types=6, mode="fast")
t_proc = ev(lambda: hd.proc_np_hv(npm_d, hv_d))
t_post = ev(lambda: model.postproc_batch(npm_d, hv_d, tp_d), reps=3)
print(f"proc_np_hv {n} x 164^2: {t_proc:.3f} ms ({n * 164 * 164 * 20 / t_proc / 1e6:.1f} GB/s algorithmic); postproc_batch incl. tables / contours / D2H: {t_post:.3f} ms")
