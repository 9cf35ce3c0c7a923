import sys
sys.path.insert(0, "/root/repo")
import torch
from tiatoolbox_amd.models.architecture.fused import hip_stem_conv_pool, pack_stem_weights
x = torch.randint(0, 255, (1024, 256, 256, 3), dtype=torch.uint8, device="cuda")
w = torch.randn(64, 3, 7, 7, device="cuda")
wp = pack_stem_weights(w)
b = torch.randn(64, device="cuda")
y = hip_stem_conv_pool(x, wp, b)
torch.cuda.synchronize()
