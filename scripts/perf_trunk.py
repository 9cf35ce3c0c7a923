"""The float32 resnet18 trunk on the hand-written kernels at the bench's micro-batch: stem kernel + the 19 block
convolutions, per-stage HIP-event times.  Also the workload of the convolution kernels' rocprofv3 passes
(`rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, one counter per run).
usage: perf_trunk.py [batch=1024] [patch=256] [reps=5] [pmc]
"pmc": the counter-pass workload -- the 19 block convolutions of exactly three forwards and nothing else (a 4096-patch layer call is
several dispatches of < 2 GiB input each, so bytes per CALL = counter total / (launches per forward x forwards); the line
"PMC forwards=3" tells the summary how many)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tiatoolbox_amd.models.architecture import get_pretrained_model
from tiatoolbox_amd.models.architecture.fused import MfmaResNet, fuse_cnn_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
model, _ = get_pretrained_model("resnet18-kather100k")
m = fuse_cnn_model(model, epilogue_fusion="mfma").cuda().to(memory_format=torch.channels_last).eval()
trunk = next(t for t in m.modules() if isinstance(t, MfmaResNet))
x = torch.randint(0, 256, (n, hw, hw, 3), dtype=torch.uint8, device="cuda")

def ev(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

if len(sys.argv) > 4 and sys.argv[4] == "pmc":
    with torch.inference_mode():
        feat = trunk.stem_forward(x)
        for _ in range(3):
            trunk.blocks(feat)
    torch.cuda.synchronize()
    print("PMC forwards=3")
    sys.exit(0)

with torch.inference_mode():
    feat = trunk.stem_forward(x)
    t_stem = ev(lambda: trunk.stem_forward(x))
    t_blocks = ev(lambda: trunk.blocks(feat))
    t_all = ev(lambda: m(x.permute(0, 3, 1, 2)))
ho = (hw - 1) // 2 + 1
f_stem = 2.0 * n * ho * ho * 64 * 147
f_all = 3.64e9 * (hw / 224.0) ** 2 * n
print(f"stem   n={n} {hw}x{hw}: {t_stem:.3f} ms  {f_stem / t_stem / 1e9:.1f} TF/s ({f_stem / t_stem / 1e9 / 157.3 * 100:.1f}% of 157.3)")
print(f"blocks n={n}: {t_blocks:.3f} ms  {(f_all - f_stem) / t_blocks / 1e9:.1f} TF/s ({(f_all - f_stem) / t_blocks / 1e9 / 157.3 * 100:.1f}%)")
print(f"forward (stem + blocks + pool + classifier + softmax): {t_all:.3f} ms  {f_all / t_all / 1e9:.1f} TF/s ({f_all / t_all / 1e9 / 157.3 * 100:.1f}%)")
