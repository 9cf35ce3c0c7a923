"""Developer A/B of the tap-reuse kernel on the 3x3 / stride-1 layer shapes of resnet18 (256^2 and 224^2 patches), hand-written
kernel only, clock warm-up, best of three interleaved rounds.  Run under the developer switches (TIA_CONV_PRIO=0..3, ...)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tiatoolbox_amd.models.architecture.fused import hip_conv2d, pack_conv_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = [(64, 64, 4), (128, 32, 3), (256, 16, 3), (512, 8, 3), (64, 56, 4), (128, 28, 3), (256, 14, 3), (512, 7, 3)]

def ev(fn, reps=10, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out, tot = [], {256: 0.0, 224: 0.0}
for c, hw, count in shapes:
    conv = torch.nn.Conv2d(c, c, 3, padding=1).cuda()
    x = torch.randn((n, c, hw, hw), device="cuda").contiguous(memory_format=torch.channels_last)
    res = torch.randn_like(x)
    wp = pack_conv_weights(conv)
    with torch.inference_mode():
        t = min(ev(lambda: hip_conv2d(x, wp, conv.bias, res, kernel=3, stride=1, padding=1, relu=True)) for _ in range(3))
    tot[256 if hw in (64, 32, 16, 8) else 224] += t * count
    out.append(f"{c}@{hw}: {t:.3f} ms {2.0 * n * hw * hw * c * c * 9 / t / 1e9:.1f}")
    del x, res
print(f"n={n}: " + " | ".join(out) + f" | 13 launches: 256^2 {tot[256]:.2f} ms, 224^2 {tot[224]:.2f} ms")
