import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from tiatoolbox_amd.tools import reinhard as rh
from tiatoolbox_amd.utils import synth
tgt = np.load(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests/golden/target_crop_256.npy'))
norm = rh.ReinhardNormalizer(); norm.fit(tgt)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blocks = synth.g_he(16, 256, 256, seed=3)
one = torch.from_numpy(np.tile(blocks[:16].reshape(4, 4, 256, 256, 3).transpose(0, 2, 1, 3, 4).reshape(1024, 1024, 3), (side // 1024, side // 1024, 1))[None]).cuda().contiguous()
for _ in range(5):
    norm.transform(one)
torch.cuda.synchronize()
