"""Developer micro-benchmark of the dictionary-learning statistics kernel (TIA_MODE_VAHADANE) alone."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.utils import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = torch.from_numpy(synth.g_he(64, hw, hw, seed=1)).cuda().repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
for mat in (False, True):
    p = dev.make_params(mode=_lib.MODE_VAHADANE, dl_one_kernel=mat)
    dev.stain_stats(x, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dev.stain_stats(x, p)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 3
    print(f"vahadane stats n={n} {hw}x{hw} one_kernel={int(mat)}: {t:.2f} ms  {n / t * 1e3:,.0f} patches/s  ({t / n * 8192:.1f} ms per 8192)")
