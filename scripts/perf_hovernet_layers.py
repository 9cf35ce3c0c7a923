"""Per-layer table of the fused HoVer-Net / UNet forward: every `_Conv` call and every fused elementwise pass timed with
HIP events (one sync per call: launch gaps are excluded, the numbers are kernel times), grouped by shape."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiatoolbox_amd.models.architecture.hovernet_fused as hf  # noqa: E402
import tiatoolbox_amd.models.architecture.unet_fused as uf  # noqa: E402
from tiatoolbox_amd.models.architecture import get_pretrained_model  # noqa: E402
from tiatoolbox_amd.utils import synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "hovernet"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = collections.OrderedDict()


def timed(label_fn, fn):
    def wrapper(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        e1.synchronize()
        key, flops, nbytes = label_fn(a, k, out)
        r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
        r[2] += flops
        r[3] += nbytes
        return out
    return wrapper


orig_conv = hf._Conv.forward


def conv_label(a, k, out):
    self, x = a[0], a[1]
    first = out[1] if isinstance(out, tuple) else out
    n, co, ho, wo = first.shape
    ci = x.shape[1]
    kind = ("mfma" if self.mfma_ok else "grouped" if self.grouped_ok else "thin" if self.thin_ok else "head" if self.head_ok else "library")
    flops = 2.0 * n * ho * wo * co * (ci // self.groups) * self.kernel * self.kernel
    nbytes = 4.0 * (x.numel() + first.numel() + (k["residual"].numel() if k.get("residual") is not None else 0))
    return (f"conv {kind:7s} {self.kernel}x{self.kernel}/{self.stride} {ci:4d}->{co:4d} out {ho}x{wo}"
            f"{' +res' if k.get('residual') is not None else ''}"), flops, nbytes


hf._Conv.forward = timed(conv_label, orig_conv)
for mod, name in ((hf, "hip_conv2d_post"), (hf, "hip_conv1x1_pre"), (hf, "hip_scale_shift_act"), (hf, "hip_scale_shift_act_view"), (hf, "hip_upsample2x_add"),
                  (uf, "hip_upsample2x_add"), (uf, "hip_stem_conv_pool")):
    def lab(a, k, out, name=name):
        first = out[1] if isinstance(out, tuple) else out
        x = a[0]
        flops = 0.0
        if name == "hip_conv2d_post":
            n, co, ho, wo = first.shape
            flops = 2.0 * n * ho * wo * co * x.shape[1] * k["kernel"] ** 2
        if name == "hip_conv1x1_pre":
            n, co, ho, wo = first.shape
            flops = 2.0 * n * ho * wo * co * x.shape[1]
        if name == "hip_stem_conv_pool":
            n, h, w, _ = x.shape
            flops = 2.0 * n * (h // 2) * (w // 2) * 64 * 147
        extra = sum(t.numel() * t.element_size() for t in (out if isinstance(out, tuple) else (out,)) if t is not None)
        return f"{name} {tuple(x.shape)} -> {tuple(first.shape)}", flops, float(x.numel() * x.element_size() + extra)
    setattr(mod, name, timed(lab, getattr(mod, name)))
# conv_with_post goes through hip_conv2d_post directly, not _Conv.forward: both are wrapped above

if which == "hovernet":
    model, _ = get_pretrained_model("hovernet_fast-pannuke")
    net = hf.FusedHoVerNet(model.eval().cuda()).cuda()
    x = torch.from_numpy(synth.g_he(batch, 256, 256, seed=5)).cuda().float().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
else:
    model, _ = get_pretrained_model("fcn_resnet50_unet-bcss")
    net = uf.FusedUNet(model.eval().cuda()).cuda()
    x = torch.from_numpy(synth.g_he(batch, 1024, 1024, seed=5)).cuda().permute(0, 3, 1, 2)
with torch.inference_mode():
    net(x)
    rows.clear()
    reps = 3
    for _ in range(reps):
        net(x)
total = sum(r[1] for r in rows.values())
print(f"{which} batch {batch}: {total / reps / batch:.3f} ms per tile (sum of timed calls), {sum(r[2] for r in rows.values()) / total / 1e9:.1f} TF/s overall")
print(f"{'calls':>5s} {'ms/fwd':>8s} {'share':>6s} {'TF/s':>7s} {'GB/s':>7s}  op")
for key, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{r[0] // reps:5d} {r[1] / reps:8.3f} {100 * r[1] / total:5.1f}% {r[2] / r[1] / 1e9:7.1f} {r[3] / r[1] / 1e6:7.0f}  {key}")
