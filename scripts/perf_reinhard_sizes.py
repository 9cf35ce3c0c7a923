import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tiatoolbox_amd.tools import reinhard as rh
from tiatoolbox_amd.utils import synth
import bench_classic as bc
target = np.load(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests/golden/target_crop_256.npy'))
norm = rh.ReinhardNormalizer(); norm.fit(target)
for (n, hw) in ((12544, 128), (4096, 224), (3136, 256), (50176, 64), (784, 512)):
    host = synth.g_he(64, hw, hw, seed=1)
    x = torch.from_numpy(host).cuda().repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
    t = bc._ev_time(lambda: norm.transform(x), 10)
    t2 = bc._ev_time(lambda: norm.lab_statistics(x), 10)
    print(f"n={n} {hw}^2: transform {t*1e3:.3f} ms  {2*x.numel()/t/1e9:.0f} GB/s   stats {t2*1e3:.3f} ms {x.numel()/t2/1e9:.0f} GB/s", flush=True)
    del x
