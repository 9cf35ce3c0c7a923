#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03q
cd $R
timeout 900 python -m pytest tests/test_stem_gpu.py tests/test_engine.py -m gpu -q 2>&1 | tail -8
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/r03q_perf_stem_h.txt
import torch, sys
sys.path.insert(0, '.')
from tiatoolbox_amd.models.architecture.fused import hip_stem_conv_pool, hip_stem_conv_pool_h, pack_stem_weights, pack_stem_weights_h
def ev(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
b = torch.randn(64, device="cuda") * 0.1
for hw in (256, 224):
    x = torch.randint(0, 256, (1024, hw, hw, 3), dtype=torch.uint8, device="cuda")
    wp, wph = pack_stem_weights(w), pack_stem_weights_h(w, torch.float16)
    fl = 2.0 * 1024 * (hw // 2) ** 2 * 64 * 147
    t32 = ev(lambda: hip_stem_conv_pool(x, wp, b, out_dtype=torch.float16))
    th = ev(lambda: hip_stem_conv_pool_h(x, wph, b, dtype=torch.float16))
    byts = x.numel() + 1024 * (hw // 4) ** 2 * 64 * 2
    print(f"stem 1024 x {hw}^2: f32 MFMA {t32:.3f} ms ({fl/t32/1e9:.1f} TF/s) | half MFMA {th:.3f} ms ({fl/th/1e9:.1f} TF/s, {byts/th/1e6:.0f} GB/s)")
PY
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['extras']['cnn_float16'])"
