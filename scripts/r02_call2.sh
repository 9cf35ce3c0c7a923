#!/bin/bash
# Round-2 GPU call 2: new GPU tests (Vahadane on device, HoVerNet+, stage planes), default bench, Vahadane timing.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/r02b_pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r02b_bench.json 2> $OUT/r02b_bench.err; echo "bench rc=$?"; tail -c 4000 $OUT/r02b_bench.json; tail -5 $OUT/r02b_bench.err
echo "== vahadane timing"
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/r02b_vahadane.txt
import time, torch, numpy as np
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.tools.stainextract import VahadaneExtractor, MacenkoExtractor
from tiatoolbox_amd.utils import synth
import logging; logging.getLogger("tiatoolbox_amd").setLevel(logging.ERROR)
for hw in (224, 256):
    host = synth.g_he(256, hw, hw, seed=1)
    x = torch.from_numpy(host).cuda().repeat(8, 1, 1, 1).contiguous()
    for name, ex in (("vahadane", VahadaneExtractor()), ("macenko", MacenkoExtractor())):
        p = ex.stats_params()
        st = dev.stain_stats(x, p); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): st = dev.stain_stats(x, p)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        it = st[:, 11].cpu().numpy() if name == "vahadane" else None
        print(f"{name} stats n={x.shape[0]} {hw}x{hw}: {dt*1e3:.3f} ms -> {x.shape[0]/dt:,.0f} patches/s", "iters:", None if it is None else np.bincount(it.astype(int)))
PY
