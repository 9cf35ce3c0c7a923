"""resnet18 BasicBlock convolutions in fp16 / bf16 at the bench's batch: the hand-written MFMA implicit GEMM (conv_mfma_h.hip,
epilogue fused) vs the library convolution (solver search on) + the separate HIP epilogue, per layer shape and for the trunk.
usage: perf_conv_h.py [batch=1024] [patch=256] [dtype=float16]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, torch.nn.functional as F
from tiatoolbox_amd.models.architecture.fused import hip_conv2d_h, pack_conv_weights_h, hip_bias_act_

torch.backends.cudnn.benchmark = True
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hw0 = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dt = getattr(torch, sys.argv[3] if len(sys.argv) > 3 else "float16")

def ev(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

q = hw0 // 4
shapes = [("layer1 3x3 64->64", 64, 64, q, 3, 1, 4), ("layer2 3x3/2 64->128", 64, 128, q, 3, 2, 1),
          ("layer2 1x1/2 64->128", 64, 128, q, 1, 2, 1), ("layer2 3x3 128->128", 128, 128, q // 2, 3, 1, 3),
          ("layer3 3x3/2 128->256", 128, 256, q // 2, 3, 2, 1), ("layer3 1x1/2 128->256", 128, 256, q // 2, 1, 2, 1),
          ("layer3 3x3 256->256", 256, 256, q // 4, 3, 1, 3), ("layer4 3x3/2 256->512", 256, 512, q // 4, 3, 2, 1),
          ("layer4 1x1/2 256->512", 256, 512, q // 4, 1, 2, 1), ("layer4 3x3 512->512", 512, 512, q // 8, 3, 1, 3)]
tot_h = tot_m = tot_f = 0.0
for name, cin, cout, hw, k, s, count in shapes:
    pad = 1 if k == 3 else 0
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=pad).cuda()
    x = torch.randn((n, cin, hw, hw), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    wp = pack_conv_weights_h(conv, dt)
    wh, bh = conv.weight.detach().to(dt).contiguous(memory_format=torch.channels_last), conv.bias.detach().to(dt)
    ho = (hw + 2 * pad - k) // s + 1
    res = torch.randn((n, cout, ho, ho), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    with torch.inference_mode():
        th = ev(lambda: hip_conv2d_h(x, wp, conv.bias.detach(), res, cout=cout, kernel=k, stride=s, padding=pad, relu=True))
        tm = ev(lambda: hip_bias_act_(F.conv2d(x, wh, None, s, pad).contiguous(memory_format=torch.channels_last), bh, res))
    fl = 2.0 * n * ho * ho * cout * cin * k * k
    tot_h += th * count; tot_m += tm * count; tot_f += fl * count
    print(f"{name:24s} @{hw:3d} hip {th:7.3f} ms {fl/th/1e9:7.1f} TF/s | library+epilogue {tm:7.3f} ms {fl/tm/1e9:7.1f} TF/s", flush=True)
print(f"trunk blocks total (n={n}, {hw0}x{hw0}, {dt}): hip {tot_h:.2f} ms {tot_f/tot_h/1e9:.1f} TF/s ({tot_f/tot_h/1e9/2500*100:.1f}% of 2500) | "
      f"library+epilogue {tot_m:.2f} ms {tot_f/tot_m/1e9:.1f} TF/s")
