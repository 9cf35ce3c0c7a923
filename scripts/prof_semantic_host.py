"""Host-side profile (cProfile) of one SemanticSegmentor WSI run on the bench's synthetic 20,000^2 slide: where the time
between the forwards goes (mask, grid, merge launches, D2H, npz write)."""
import cProfile, os, pstats, shutil, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tiatoolbox_amd.models.engine.semantic_segmentor import SemanticSegmentor
from tiatoolbox_amd.utils import synth
from tiatoolbox_amd.wsicore import ArrayWSIReader

side = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
device = torch.device("cuda")
tile = torch.from_numpy(synth.g_he(1, 2048, 2048, seed=3)[0]).to(device)
slide = torch.full((side, side, 3), 243, dtype=torch.uint8, device=device)
lo, hi = side // 10, side - side // 10
for y in range(lo, hi, 2048 + 256):
    for x in range(lo, hi, 2048 + 256):
        h, w = min(2048, hi - y), min(2048, hi - x)
        slide[y:y + h, x:x + w] = tile[:h, :w]
reader = ArrayWSIReader(slide, mpp=0.25, power=40.0)
eng = SemanticSegmentor("fcn_resnet50_unet-bcss", batch_size=8, device="cuda", verbose=False)
scratch = Path(tempfile.mkdtemp(prefix="tia_sem_", dir="/dev/shm"))
eng.run([reader], patch_mode=False, save_dir=scratch / "out", overwrite=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
eng.run([reader], patch_mode=False, save_dir=scratch / "out", overwrite=True)
torch.cuda.synchronize()
pr.disable()
print(f"run: {time.perf_counter() - t0:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
shutil.rmtree(scratch, ignore_errors=True)
