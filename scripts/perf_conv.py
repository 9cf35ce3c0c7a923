"""resnet18 BasicBlock convolutions at the bench's batch: the hand-written MFMA implicit GEMM (conv_mfma.hip, epilogue
fused) vs MIOpen (solver search on) + the separate HIP epilogue, per layer shape and for the whole trunk.  fp32."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, torch.nn.functional as F
from tiatoolbox_amd.models.architecture.fused import (hip_bias_act_, hip_conv2d, hip_conv3x3_wino, pack_conv_weights,
                                                      pack_conv_weights_wino)

torch.backends.cudnn.benchmark = True
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024

def ev(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

shapes = [("layer1 3x3 64->64 @56", 64, 64, 56, 3, 1, 4), ("layer2 3x3/2 64->128 @56", 64, 128, 56, 3, 2, 1),
          ("layer2 1x1/2 64->128", 64, 128, 56, 1, 2, 1), ("layer2 3x3 128->128 @28", 128, 128, 28, 3, 1, 3),
          ("layer3 3x3/2 128->256 @28", 128, 256, 28, 3, 2, 1), ("layer3 1x1/2 128->256", 128, 256, 28, 1, 2, 1),
          ("layer3 3x3 256->256 @14", 256, 256, 14, 3, 1, 3), ("layer4 3x3/2 256->512 @14", 256, 512, 14, 3, 2, 1),
          ("layer4 1x1/2 256->512", 256, 512, 14, 1, 2, 1), ("layer4 3x3 512->512 @7", 512, 512, 7, 3, 1, 3)]
tot_h = tot_m = tot_f = tot_w = 0.0
for name, cin, cout, hw, k, s, count in shapes:
    pad = 1 if k == 3 else 0
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=pad).cuda()
    x = torch.randn((n, cin, hw, hw), device="cuda").contiguous(memory_format=torch.channels_last)
    wp = pack_conv_weights(conv)
    ho = (hw + 2 * pad - k) // s + 1
    res = torch.randn((n, cout, ho, ho), device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.inference_mode():
        th = ev(lambda: hip_conv2d(x, wp, conv.bias, res, kernel=k, stride=s, padding=pad, relu=True))
        tm = ev(lambda: hip_bias_act_(F.conv2d(x, conv.weight, None, s, pad).contiguous(memory_format=torch.channels_last), conv.bias, res))
        tw = None
        if k == 3 and s == 1:  # the opt-in Winograd form of the same layer (conv_algo="winograd")
            up = pack_conv_weights_wino(conv)
            tw = ev(lambda: hip_conv3x3_wino(x, up, conv.bias, res, padding=pad, relu=True))
    fl = 2.0 * n * ho * ho * cout * cin * k * k
    tot_h += th * count; tot_m += tm * count; tot_f += fl * count
    tot_w += (tw if tw is not None else th) * count
    wino = f" | hip winograd {tw:7.3f} ms {fl/tw/1e9:7.1f} TF/s effective" if tw is not None else ""
    print(f"{name:28s} hip {th:7.3f} ms {fl/th/1e9:7.1f} TF/s | miopen+epilogue {tm:7.3f} ms {fl/tm/1e9:7.1f} TF/s{wino}", flush=True)
print(f"trunk blocks total (n={n}): hip {tot_h:.2f} ms {tot_f/tot_h/1e9:.1f} TF/s ({tot_f/tot_h/1e9/157.3*100:.1f}% of 157.3) | "
      f"miopen+epilogue {tot_m:.2f} ms {tot_f/tot_m/1e9:.1f} TF/s | hip with conv_algo='winograd' {tot_w:.2f} ms {tot_f/tot_w/1e9:.1f} TF/s effective")
