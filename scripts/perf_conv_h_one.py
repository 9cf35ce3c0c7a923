"""One half-precision convolution shape in a loop (workload of the rocprofv3 --pmc passes of conv_mfma_h_kernel).
usage: perf_conv_h_one.py [n=1024] [cin=128] [cout=128] [hw=32] [k=3] [stride=1] [reps=20]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tiatoolbox_amd.models.architecture.fused import hip_conv2d_h, pack_conv_weights_h

a = [int(v) for v in sys.argv[1:]] + [1024, 128, 128, 32, 3, 1, 20][len(sys.argv) - 1:]
n, cin, cout, hw, k, s, reps = a
pad = 1 if k == 3 else 0
conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=pad).cuda()
x = torch.randn((n, cin, hw, hw), device="cuda").half().contiguous(memory_format=torch.channels_last)
wp = pack_conv_weights_h(conv, torch.float16)
ho = (hw + 2 * pad - k) // s + 1
res = torch.randn((n, cout, ho, ho), device="cuda").half().contiguous(memory_format=torch.channels_last)
for _ in range(3):
    hip_conv2d_h(x, wp, conv.bias.detach(), res, cout=cout, kernel=k, stride=s, padding=pad, relu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    hip_conv2d_h(x, wp, conv.bias.detach(), res, cout=cout, kernel=k, stride=s, padding=pad, relu=True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps
print(f"{cin}->{cout} k{k}/{s} @{hw} n={n}: {t:.3f} ms {2.0*n*ho*ho*cout*cin*k*k/t/1e9:.1f} TF/s")
