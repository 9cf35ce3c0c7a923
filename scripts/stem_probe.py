"""One launch of the stem kernel on 1024 uint8 patches of 256 x 256 -- with a timing build of the library (TIA_LIB_PATH, built with
build.build(defines=("TIA_STEM_TIMING=1",), out=...)) it prints the per-phase cycles and the sustained shader clock."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tiatoolbox_amd.models.architecture.fused import hip_stem_conv_pool, pack_stem_weights
x = torch.randint(0, 255, (1024, 256, 256, 3), dtype=torch.uint8, device="cuda")
w = torch.randn(64, 3, 7, 7, device="cuda")
wp = pack_stem_weights(w)
b = torch.randn(64, device="cuda")
y = hip_stem_conv_pool(x, wp, b)
torch.cuda.synchronize()
