"""Workloads of the per-kernel ``rocprofv3 --pmc FETCH_SIZE`` / ``WRITE_SIZE`` passes behind ``roofline.traffic`` of the
non-headline bench lines: exactly the shapes ``bench_configs.py`` times.  Each call of the measured operation is bracketed by a
marker kernel (``torch.zeros(7)``: one ``fill`` launch) so that per-CALL totals can be formed from per-dispatch counter means.

usage: pmc_workloads.py vahadane|hover|canvas [calls=3]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

which = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev_ = torch.device("cuda", 0)

if which == "vahadane":
    from tiatoolbox_amd import _lib
    from tiatoolbox_amd.tools import _stain_device as dev
    from tiatoolbox_amd.utils import synth

    n, hw = 8192, 256
    x = torch.from_numpy(synth.g_he(256, hw, hw, seed=11)).to(dev_).repeat(n // 256, 1, 1, 1).contiguous()
    target = np.array([[0.55, 0.76, 0.35], [0.1, 0.96, 0.27]])
    p = dev.make_params(mode=_lib.MODE_VAHADANE, target_stain=target, target_maxc=np.array([[1.9, 1.0]]))
    for _ in range(calls):
        dev.stain_stats(x, p)
elif which == "hover":
    from tiatoolbox_amd.utils import synth
    from tiatoolbox_amd.models.architecture import _hover_device as hd

    npm, hv, _ = synth.hover_head_maps(8, 164, 164, seed=1, n_blobs=60)
    npm_d = torch.from_numpy(npm).to(dev_).repeat(32, 1, 1, 1)
    hv_d = torch.from_numpy(hv).to(dev_).repeat(32, 1, 1, 1)
    for _ in range(calls):
        hd.proc_np_hv(npm_d, hv_d)
elif which == "canvas":
    from tiatoolbox_amd.models.engine.semantic_segmentor import _finalize, _row_merge
    from tiatoolbox_amd.tools.patchextraction import PatchExtractor
    from tiatoolbox_amd.wsicore import ArrayWSIReader

    side, oh_, ph = 20000, 512, 1024
    in_b, out_b = PatchExtractor.get_coordinates(patch_output_shape=(oh_, oh_), image_shape=(side, side), patch_input_shape=(ph, ph),
                                                 stride_shape=(450, 450))
    sel = out_b[:, 1] == out_b[0, 1]
    blocks = torch.rand((int(sel.sum()), oh_, oh_, 5), device=dev_)
    xs = out_b[sel][:, 0]
    row, cnt = _row_merge(blocks, xs, side)
    row2, cnt2 = row.clone(), cnt.clone()
    band = torch.zeros((oh_, side), dtype=torch.uint8, device=dev_)
    slide = torch.randint(0, 255, (4096, 4096, 3), dtype=torch.uint8)
    reader = ArrayWSIReader(slide.numpy(), mpp=0.25, power=40.0)
    bounds = np.array([[x0, y0, x0 + ph, y0 + ph] for x0, y0 in ((0, 0), (900, 0), (1800, 0), (2700, 0), (0, 900), (900, 900), (1800, 900), (2700, 900))])
    for _ in range(calls):
        _row_merge(blocks, xs, side)
        _finalize(row, cnt, 0, row2, cnt2, 450, 450, 900, None, band, y_base=450)
        reader.read_bounds_batch(bounds)
else:
    raise SystemExit(f"unknown workload {which}")
torch.cuda.synchronize()
print(f"{which}: {calls} calls done")
