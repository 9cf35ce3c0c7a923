"""Developer benchmark: resnet18 forward variants on one MI355X."""
import sys, time, logging
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
logging.getLogger("tiatoolbox_amd").setLevel(logging.ERROR)
from tiatoolbox_amd.models.architecture import get_pretrained_model
from tiatoolbox_amd.models.architecture.fused import fuse_cnn_model

def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

m, _ = get_pretrained_model("resnet18-kather100k"); m.eval()
n = 4096
for dtype in (torch.float16, torch.bfloat16, torch.float32):
    x = torch.rand(n, 224, 224, 3, device="cuda").to(dtype)
    variants = {"plain": m, "folded": fuse_cnn_model(m, epilogue_fusion=False), "mfma": fuse_cnn_model(m, epilogue_fusion="mfma")}
    ref = None
    for name, mod in variants.items():
        mod = mod.to("cuda").to(dtype).to(memory_format=torch.channels_last).eval()
        for mb in (512, 1024):
            def run():
                with torch.inference_mode():
                    return [mod(x[s:s+mb].permute(0,3,1,2)) for s in range(0, n, mb)]
            try:
                t = timeit(run)
                out = torch.cat(run())
                if ref is None: ref = out
                err = (out - ref).abs().max().item()
                print(f"{str(dtype):14s} {name:7s} mb={mb:5d}: {t:8.2f} ms  {n/t*1e3:10,.0f} patches/s  {3.64e9*n/t/1e9:7.1f} TF/s  maxdiff {err:.2e}", flush=True)
            except Exception as e:
                print(name, mb, "FAILED", repr(e)[:200], flush=True)
