"""Developer micro-benchmark of the stain kernels (HIP-event timing on the launch stream)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.tools.stainnorm import get_normalizer
from tiatoolbox_amd.utils import synth

def timeit(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

n, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 224, 224
if len(sys.argv) > 2: h = w = int(sys.argv[2])
base = torch.from_numpy(synth.g_he(64, h, w, seed=1)).cuda()
x = base.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
norm = get_normalizer("macenko"); norm.fit(x[0])
p = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
stats = dev.stain_stats(x, p)
t = timeit(lambda: dev.stain_stats(x, p))
print(f"stats   n={n} {h}x{w}: {t:.3f} ms  -> {n/t*1e3:,.0f} patches/s, {t/n*1e3:.2f} us/patch; "
      f"{n*h*w*3/t/1e6:.1f} GB/s ({n*h*w*3/t/1e6/80:.2f}% of 8 TB/s); patches handed back to the streaming kernel: "
      f"{dev.redo_count(x.device, n, h, w)}")
ps = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target); ps.select_mode = 1
ts = timeit(lambda: dev.stain_stats(x, ps), reps=5)
print(f"stats (streaming kernel only, select_mode=1): {ts:.3f} ms")
cyc = dev.stain_stats(x, p)[:, _lib.ST_CYCLES:_lib.ST_CYCLES+16].mean(0).cpu().numpy()
names = ["P1","LUT","P2","EIG","SEL_HIST","SEL_FIND","SEL_COLLECT","SEL_SORT","PHI_TOTAL","CONC_TOTAL","TOTAL","clv0","clv1","ccnt0","ccnt1","philv"]
print("  cycles/patch:", ", ".join(f"{k}={v:,.0f}" for k, v in zip(names, cyc)))
pr = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
pr.mode = _lib.MODE_FIXED; pr.stain_fixed[:] = [0.65,0.70,0.29,0.07,0.99,0.11]
t = timeit(lambda: dev.stain_stats(x, pr))
print(f"stats(fixed S) : {t:.3f} ms  -> {n/t*1e3:,.0f} patches/s")
byts = 2 * n * h * w * 3
for math, mname in ((_lib.MATH_F32, "f32"), (_lib.MATH_F64, "f64"), (_lib.MATH_F64_REF, "f64_ref(libm exp)")):
    for ok, oname in ((_lib.OUT_U8, "u8"), (_lib.OUT_UNIT_F16, "unit_f16")):
        out = dev.stain_apply(x, stats, norm.stain_matrix_target, out_kind=ok, math=math)
        t = timeit(lambda: dev.stain_apply(x, stats, norm.stain_matrix_target, out_kind=ok, math=math, out=out))
        b = n*h*w*3*(1 + out.element_size())
        print(f"apply {mname}->{oname}: {t:.3f} ms  {b/t/1e6:.1f} GB/s ({b/t/1e6/8000*100:.1f}% of 8 TB/s)  {n/t*1e3:,.0f} patches/s")
