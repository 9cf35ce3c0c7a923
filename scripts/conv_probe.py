"""Single-shape probe of the float32 convolution entry point: `conv_probe.py n,cin,cout,hw[,k[,stride]] ...` prints the launch
time, the algorithmic TFLOP/s and the block geometry the dispatcher picked (tia_conv3x3_geometry)."""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.models.architecture.fused import hip_conv2d, pack_conv_weights


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for spec in sys.argv[1:]:
    v = [int(t) for t in spec.split(",")]
    n, cin, cout, hw = v[:4]
    k = v[4] if len(v) > 4 else 3
    s = v[5] if len(v) > 5 else 1
    pad = 1 if k == 3 else 0
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=pad).cuda()
    x = torch.randn((n, cin, hw, hw), device="cuda").contiguous(memory_format=torch.channels_last)
    wp = pack_conv_weights(conv)
    ho = (hw + 2 * pad - k) // s + 1
    res = torch.randn((n, cout, ho, ho), device="cuda").contiguous(memory_format=torch.channels_last)
    geom = (ctypes.c_int32 * 4)()
    kind = _lib.load().tia_conv3x3_geometry(hw, hw, ho, ho, pad, pad, geom) if (k == 3 and s == 1) else -1
    with torch.inference_mode():
        t = ev(lambda: hip_conv2d(x, wp, conv.bias, res, kernel=k, stride=s, padding=pad, relu=True))
    fl = 2.0 * n * ho * ho * cout * cin * k * k
    print(f"n={n} {cin}->{cout} @{hw} k{k}/{s}: {t:.3f} ms  {fl / t / 1e9:.1f} TF/s  geometry {kind} {list(geom)}", flush=True)
