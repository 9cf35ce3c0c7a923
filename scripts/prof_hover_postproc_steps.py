"""Wall-clock of each step of HoVerNet.postproc_batch on the bench's synthetic head maps (host time incl. syncs and D2H)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from tiatoolbox_amd.utils import synth
from tiatoolbox_amd.models.architecture import _hover_device as hd

n = 256
npm, hv, tp = synth.hover_head_maps(8, 164, 164, seed=1, n_blobs=60)
dev = torch.device("cuda")
np_map = torch.from_numpy(npm).to(dev).repeat(n // 8, 1, 1, 1)
hv_map = torch.from_numpy(hv).to(dev).repeat(n // 8, 1, 1, 1)
tp_map = torch.from_numpy(tp).to(dev).repeat(n // 8, 1, 1, 1)
def run(show):
    t = [time.perf_counter()]
    def lap(name):
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if show: print(f"{name:34s} {1e3 * (t[-1] - t[-2]):8.3f} ms")
    inst, nmark = hd.proc_np_hv(np_map, hv_map); lap("proc_np_hv")
    tmap = torch.round(tp_map).to(torch.uint8).reshape(inst.shape); lap("round/to uint8")
    num_types = max(6, int(tmap.max()) + 1); max_inst = int(nmark.max()); lap("max() syncs")
    stats, types = hd.instance_stats(inst, tmap, max_inst, num_types); lap("instance_stats")
    meta, points = hd.contours(inst, stats, max_inst); lap("contours (scan, write, D2H)")
    stats_h = stats.cpu().numpy(); types_h = types.cpu().numpy(); lap("stats / types D2H")
    inst_h = inst.cpu().numpy(); lap("label maps D2H")
    tables = hd.tables_from_stats_batch(stats_h, types_h, meta=meta, points=points); lap("tables_from_stats_batch")
    if show: print(f"{'total':34s} {1e3 * (t[-1] - t[0]):8.3f} ms; max_inst {max_inst}, points {len(points)}")
run(False); run(True)
