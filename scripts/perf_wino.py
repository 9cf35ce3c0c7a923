"""Winograd F(2x2, 3x3) vs the direct tap-reuse kernel, per 3x3 / stride-1 layer of resnet18 at a batch and patch size, and the whole
trunk.  usage: perf_wino.py [batch=1024] [patch=256]  (HIP events on the launch stream; TFLOP/s are DIRECT-convolution flops / time,
i.e. 'effective' for the Winograd rows, whose executed MFMA flops are 16/36 of that -- printed as `exec`)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from tiatoolbox_amd.models.architecture.fused import hip_conv2d, hip_conv3x3_wino, pack_conv_weights, pack_conv_weights_wino


def ev(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    patch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    pmc = len(sys.argv) > 3 and sys.argv[3] == "pmc"
    g = torch.Generator(device="cuda").manual_seed(0)
    tot_d = tot_w = 0.0
    for c, div, count in ((64, 4, 4), (128, 8, 3), (256, 16, 3), (512, 32, 3)):
        hw = patch // div
        conv = torch.nn.Conv2d(c, c, 3, padding=1).cuda()
        x = torch.randn((n, c, hw, hw), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        res = torch.randn_like(x)
        wp, up = pack_conv_weights(conv), pack_conv_weights_wino(conv)
        flops = 2.0 * n * hw * hw * c * c * 9
        f0 = lambda: hip_conv2d(x, wp, conv.bias, res, kernel=3, stride=1, padding=1, relu=True)  # noqa: E731
        f2 = lambda: hip_conv3x3_wino(x, up, conv.bias, res, padding=1, relu=True)  # noqa: E731
        if pmc:  # counter-pass workload: the 13 Winograd launches of exactly two forwards (see perf_trunk.py "pmc")
            for _ in range(2 * count):
                f2()
            continue
        for _ in range(30):  # the clocks ramp up over the first tens of milliseconds of load: an unwarmed first column reads 5-10 % slow
            f0()             # (the "direct" column of profiles/r05b..r05o_perf_wino*.txt was measured without this and is pessimistic)
        td, tw = ev(f0), ev(f2)
        for _ in range(2):  # interleaved rounds, best of three
            td, tw = min(td, ev(f0)), min(tw, ev(f2))
        a = hip_conv2d(x, wp, conv.bias, res, kernel=3, stride=1, padding=1, relu=False)
        b = hip_conv3x3_wino(x, up, conv.bias, res, padding=1, relu=False)
        rel = ((a - b).abs().max() / a.abs().max()).item()
        tot_d += td * count
        tot_w += tw * count
        print(f"3x3 {c:3d}->{c:3d} @{hw:3d} n={n}: direct {td:6.3f} ms {flops / td / 1e9:6.1f} TF/s | winograd {tw:6.3f} ms "
              f"{flops / tw / 1e9:6.1f} TF/s effective, {flops * 16 / 36 / tw / 1e9:6.1f} exec | x{td / tw:4.2f} | max rel diff {rel:.1e}",
              flush=True)
    if pmc:
        torch.cuda.synchronize()
        print("PMC forwards=2")
        return
    print(f"13 stride-1 3x3 launches of one resnet18 forward: direct {tot_d:.2f} ms, winograd {tot_w:.2f} ms (x{tot_d / tot_w:.2f})")


if __name__ == "__main__":
    main()
