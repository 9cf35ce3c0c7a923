"""Developer A/B of the tap-reuse kernel's band geometries: the four 3x3 / stride-1 layer shapes of resnet18 at 224^2 patches
(56 / 28 / 14 / 7 maps), hand-written kernel only, with a clock warm-up.  Run under TIA_DEV=1 and TIA_CONV_BAND_GAPS=1 (round-4 bands with the
zero rows among the GEMM rows), TIA_CONV_BAND_MAX_STRIPS=k, TIA_CONV_NO_BAND=1 (slice / ring kernels) to compare."""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tiatoolbox_amd import _lib
from tiatoolbox_amd.models.architecture.fused import hip_conv2d, pack_conv_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)]
if len(sys.argv) > 2:
    shapes = [s for s in shapes if s[2] in [int(v) for v in sys.argv[2].split(",")]]

def ev(fn, reps=20, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out = []
for cin, cout, hw in shapes:
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    x = torch.randn((n, cin, hw, hw), device="cuda").contiguous(memory_format=torch.channels_last)
    res = torch.randn((n, cout, hw, hw), device="cuda").contiguous(memory_format=torch.channels_last)
    wp = pack_conv_weights(conv)
    geom = (ctypes.c_int32 * 4)()
    kind = _lib.load().tia_conv3x3_geometry(hw, hw, hw, hw, 1, 1, geom)
    route = _lib.load().tia_conv2d_route_f32(n, hw, hw, cin, cout, 3, 3, 1, 1, 1, hw, hw)
    with torch.inference_mode():
        t = min(ev(lambda: hip_conv2d(x, wp, conv.bias, res, kernel=3, stride=1, padding=1, relu=True)) for _ in range(2))
    fl = 2.0 * n * hw * hw * cin * cout * 9
    out.append(f"{cin}->{cout} @{hw}: route {route} geometry {kind} {list(geom)}  {t:.3f} ms  {fl / t / 1e9:.1f} TF/s")
print(f"n={n}: " + " | ".join(out))
