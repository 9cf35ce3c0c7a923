"""Statistics + transform of ONE large image (4096 x 4096 and friends) against the same pixel count as a batch of 256 x 256 patches."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import bench_classic as bc
from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.tools import reinhard as rh
from tiatoolbox_amd.tools.stainnorm import get_normalizer
from tiatoolbox_amd.tools.tissuemask import OtsuTissueMasker
from tiatoolbox_amd.utils import synth

tgt = np.load(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests/golden/target_crop_256.npy'))
norm = get_normalizer("macenko"); norm.fit(tgt)
p = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
rn = rh.ReinhardNormalizer(); rn.fit(tgt)
for side in (1024, 2048, 4096, 8192):
    blocks = synth.g_he(16, 256, 256, seed=3)
    k = side // 256
    one = torch.from_numpy(np.tile(blocks[:16].reshape(4, 4, 256, 256, 3).transpose(0, 2, 1, 3, 4).reshape(1024, 1024, 3), (side // 1024, side // 1024, 1))[None]).cuda().contiguous()
    batch = torch.from_numpy(blocks).cuda().repeat(k * k // 16, 1, 1, 1).contiguous()
    t_one = bc._ev_time(lambda: dev.stain_stats(one, p), 5, 2)
    t_batch = bc._ev_time(lambda: dev.stain_stats(batch, p), 5, 2)
    st = dev.stain_stats(one, p)
    t_apply = bc._ev_time(lambda: dev.stain_apply(one, st, norm.stain_matrix_target), 5, 2)
    t_rh = bc._ev_time(lambda: rn.transform(one), 5, 2)
    om = OtsuTissueMasker()
    t_otsu = bc._ev_time(lambda: om.fit(one), 5, 2)
    print(f"{side}^2: macenko stats one image {t_one*1e3:.3f} ms | {k*k} patches of 256^2 {t_batch*1e3:.3f} ms | ratio {t_one/t_batch:.2f} | apply {t_apply*1e3:.3f} ms | "
          f"reinhard.transform {t_rh*1e3:.3f} ms | otsu.fit {t_otsu*1e3:.3f} ms", flush=True)
    print("   diag (fallback, list sizes) angles:", st[0, 48:53].cpu().tolist(), "conc:", st[0, 53:58].cpu().tolist(), flush=True)
