"""HoVer-Net post-processing: oracle vs the real reference (CPU), HIP vs oracle (GPU, bit-exact labels)."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest

from oracle import hovernet as oh

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "hover_golden.npz")


def _check_polys(info: dict, gold, tag: str, i: int) -> None:
    polys = [v["contours"] for v in info.values()]
    assert all(c.dtype == np.int32 and c.ndim == 2 and c.shape[1] == 2 for c in polys)
    assert np.array_equal(np.array([len(c) for c in polys]), gold[f"{tag}_polylen{i}"])
    assert np.array_equal(np.concatenate(polys), gold[f"{tag}_poly{i}"])


def _maps(gold, tag):
    h, w, seed, nb = (int(v) for v in gold[f"{tag}_shape"])
    return oh.synth_maps(2, h, w, seed=seed, n_blobs=nb)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_real_reference(gold, tag):
    npm, hv, tp = _maps(gold, tag)
    for i in range(2):
        inst = oh.proc_np_hv(npm[i], hv[i])
        assert np.array_equal(inst, gold[f"{tag}_inst"][i])
        info = oh.get_instance_info(inst, np.around(tp[i]).astype("uint8")[..., 0])
        assert np.array_equal(np.array(list(info)), gold[f"{tag}_ids{i}"])
        assert np.array_equal(np.array([v["box"] for v in info.values()]), gold[f"{tag}_box{i}"])
        np.testing.assert_array_equal(np.array([v["centroid"] for v in info.values()]), gold[f"{tag}_cent{i}"])
        assert np.array_equal(np.array([v["type"] for v in info.values()]), gold[f"{tag}_type{i}"])
        np.testing.assert_array_equal(np.array([v["prob"] for v in info.values()]), gold[f"{tag}_prob{i}"])
        _check_polys(info, gold, tag, i)


def test_find_contours_known_answers():
    """Structural known answers of ``cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE)`` (cv2 itself is absent:
    parity unpinned).  Rectangle order = the reference's own fixture ``tests/test_utils.py:2299``."""
    from oracle import cvref

    m = np.zeros((30, 30), np.uint8)
    m[10:21, 10:21] = 1
    assert cvref.first_contour(m).tolist() == [[10, 10], [10, 20], [20, 20], [20, 10]]  # TL, BL, BR, TR
    m[13:17, 14:18] = 0  # a hole does not change element 0 and is the outer border's child
    tree = cvref.find_contours_tree(m)
    assert [(b["is_hole"], b["parent"]) for b in tree] == [(False, -1), (True, 0)]
    assert cvref.first_contour(m).tolist() == [[10, 10], [10, 20], [20, 20], [20, 10]]
    assert tree[1]["points"].tolist() == [[13, 13], [14, 12], [17, 12], [18, 13], [18, 16], [17, 17], [14, 17], [13, 16]]
    d = np.zeros((7, 7), np.uint8)
    for r in range(7):
        d[r, abs(r - 3):7 - abs(r - 3)] = 1
    assert cvref.first_contour(d).tolist() == [[3, 0], [0, 3], [3, 6], [6, 3]]      # diamond: 4 vertices
    line = np.zeros((3, 8), np.uint8)
    line[1, 1:7] = 1
    assert cvref.first_contour(line).tolist() == [[1, 1], [6, 1]]                   # < 3 vertices: dropped upstream
    two = np.zeros((6, 12), np.uint8)
    two[1:3, 1:4] = 1
    two[3:5, 7:10] = 1
    assert cvref.first_contour(two).tolist() == [[7, 3], [7, 4], [9, 4], [9, 3]]    # the component found last
    # every vertex lies on the component and the APPROX_NONE chain is a closed 8-connected walk
    rng = np.random.default_rng(0)
    for _ in range(50):
        r = (rng.random((12, 12)) < 0.6).astype(np.uint8)
        if not r.any():
            continue
        pts = cvref.first_contour(r)
        assert all(r[y, x] for x, y in pts)
    info = oh.get_instance_info(np.pad(line.astype(np.int32), 2))
    assert info == {}  # hovernet.py:695-699


def test_sobel_kernels_known_values():
    from oracle import cvref

    kx, ky = cvref.sobel_kernels(5, 1, 0)
    assert kx.tolist() == [-1, -2, 0, 2, 1] and ky.tolist() == [1, 4, 6, 4, 1]   # OpenCV docs: 5x5 Sobel
    kx, ky = cvref.sobel_kernels(21, 1, 0)
    assert ky[10] == 184756 and kx[0] == -1 and kx[-1] == 1 and kx[10] == 0      # C(20,10); antisymmetric
    assert cvref.get_structuring_element_ellipse((5, 5)).tolist() == [
        [0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_proc_np_hv_bit_exact(gold, tag):
    """Instance label maps must equal the reference's (golden) and the oracle's, pixel for pixel."""
    import torch

    from tiatoolbox_amd.models.architecture import _hover_device as hd

    npm, hv, tp = _maps(gold, tag)
    inst, nmark = hd.proc_np_hv(torch.from_numpy(npm).cuda(), torch.from_numpy(hv).cuda())
    got = inst.cpu().numpy()
    assert got.dtype == np.int32
    for i in range(2):
        assert np.array_equal(got[i], gold[f"{tag}_inst"][i]), f"plane {i}: {(got[i] != gold[f'{tag}_inst'][i]).sum()} px differ"
        assert int(nmark[i]) >= got[i].max()
    # instance info
    pred_type = torch.from_numpy(np.around(tp).astype("uint8")[..., 0]).cuda()
    stats, types = hd.instance_stats(inst, pred_type, int(nmark.max()), num_types=8)
    meta, points = hd.contours(inst, stats, int(nmark.max()))
    for i in range(2):
        info = hd.info_from_stats(stats[i].cpu().numpy(), types[i].cpu().numpy(), meta=meta[i], points=points)
        _check_polys(info, gold, tag, i)
        assert np.array_equal(np.array(list(info)), gold[f"{tag}_ids{i}"])
        assert np.array_equal(np.array([v["box"] for v in info.values()]), gold[f"{tag}_box{i}"])
        np.testing.assert_array_equal(np.array([v["centroid"] for v in info.values()]), gold[f"{tag}_cent{i}"])
        assert np.array_equal(np.array([v["type"] for v in info.values()]), gold[f"{tag}_type{i}"])
        np.testing.assert_array_equal(np.array([v["prob"] for v in info.values()]), gold[f"{tag}_prob{i}"])


@pytest.mark.gpu
def test_hip_proc_np_hv_vs_oracle_more_shapes():
    import torch

    from tiatoolbox_amd.models.architecture import _hover_device as hd

    # 164^2 and smaller: the six-launch tile-resident path; 256^2: the multi-launch path; 190^2 (> 32,000 pixels: labelling in
    # LDS, marker pipeline multi-launch), 181 x 176 (just under the limit), 40 x 700 (rows too wide for the Sobel bands): the mixes
    for (h, w, seed, nb) in ((64, 80, 5, 6), (164, 164, 6, 60), (256, 256, 7, 90), (33, 47, 8, 3), (190, 190, 9, 70), (181, 176, 10, 60),
                             (40, 700, 11, 40)):
        npm, hv, _ = oh.synth_maps(3, h, w, seed=seed, n_blobs=nb)
        inst, _ = hd.proc_np_hv(torch.from_numpy(npm).cuda(), torch.from_numpy(hv).cuda())
        got = inst.cpu().numpy()
        for i in range(3):
            exp = oh.proc_np_hv(npm[i], hv[i])
            assert np.array_equal(got[i], exp), (h, w, i, int((got[i] != exp).sum()))
    # empty / full planes
    z = torch.zeros((2, 40, 40, 1), device="cuda")
    inst, n = hd.proc_np_hv(z, torch.zeros((2, 40, 40, 2), device="cuda"))
    assert int(inst.abs().sum()) == 0 and int(n.sum()) == 0
    one = torch.ones((1, 40, 40, 1), device="cuda")
    hvr = torch.rand((1, 40, 40, 2), device="cuda", generator=torch.Generator("cuda").manual_seed(0))
    inst, _ = hd.proc_np_hv(one, hvr)
    exp = oh.proc_np_hv(one[0].cpu().numpy(), hvr[0].cpu().numpy())
    assert np.array_equal(inst[0].cpu().numpy(), exp)


def _tie_cases():
    """Watershed inputs whose queue entries tie on (value, age) or on value: the pop order then depends on the heap's
    internal arrangement (skimage ``heap_general.pxi``), which is what the GPU flood must reproduce."""
    rng = np.random.default_rng(17)
    cases = []
    # (a) exactly flat plateau touched by two / many single-pixel markers (all markers carry age 0)
    for h, w, k in ((9, 13, 2), (24, 31, 5), (40, 40, 23), (64, 64, 150), (7, 200, 40)):
        img = np.zeros((h, w))
        mk = np.zeros((h, w), np.int32)
        pos = rng.choice(h * w, k, replace=False)
        mk.ravel()[pos] = np.arange(1, k + 1)
        cases.append((img, mk, np.ones((h, w), bool)))
    # (b) multi-pixel markers on a plateau (many age-0 entries per label), mask with holes -> several blobs
    img = np.zeros((50, 70))
    mk = np.zeros((50, 70), np.int32)
    mk[5:9, 5:12] = 1
    mk[30:33, 40:60] = 2
    mk[20, 20] = 3
    mk[45:48, 3:6] = 4
    mask = rng.random((50, 70)) < 0.93
    cases.append((img, mk, mask))
    # (c) staircases: few distinct levels, random markers, random masks
    for levels, shape, k in ((2, (33, 47), 7), (3, (60, 60), 30), (4, (96, 120), 80), (8, (164, 164), 200)):
        img = rng.integers(0, levels, shape).astype(np.float64)
        mk = np.zeros(shape, np.int32)
        pos = rng.choice(shape[0] * shape[1], k, replace=False)
        mk.ravel()[pos] = rng.permutation(k) + 1
        mask = rng.random(shape) < 0.9
        cases.append((img, mk, mask))
    # (d) terraces: smooth ramp quantised to steps, markers as blobs of equal value
    yy, xx = np.mgrid[0:80, 0:100]
    img = np.floor(((yy - 40) ** 2 + (xx - 50) ** 2) / 150.0)
    mk = np.zeros((80, 100), np.int32)
    mk[38:43, 20:25] = 1
    mk[38:43, 75:80] = 2
    mk[10:12, 48:53] = 3
    cases.append((img, mk, np.ones((80, 100), bool)))
    # (e) negative zero / equal negative values as produced by dist = -GaussianBlur(...)
    img = -np.round(rng.random((40, 44)), 1)
    mk = np.zeros((40, 44), np.int32)
    mk.ravel()[rng.choice(40 * 44, 25, replace=False)] = np.arange(1, 26)
    cases.append((img, mk, rng.random((40, 44)) < 0.95))
    return cases


def test_tie_cases_discriminate_heap_procedures():
    """The forced-tie inputs are meaningful: on at least one of them a heapq-style pop (bubble to a leaf, sift back)
    produces a different label map from skimage's procedure -- so a GPU flood passing them implements the latter."""
    import heapq

    from oracle import skref

    def watershed_heapq(image, markers, mask):
        h, w = image.shape
        out = np.where(mask, markers, 0).astype(np.int32).ravel()
        flat, fm = image.ravel(), np.asarray(mask, bool).ravel()
        heap, age = [], 0
        for idx in np.flatnonzero(out):
            heapq.heappush(heap, _Keyed(float(flat[idx]), 0, int(idx)))
        while heap:
            idx = heapq.heappop(heap).index
            r, c = divmod(idx, w)
            for dr, dc in ((-1, 0), (0, -1), (0, 1), (1, 0)):
                rr, cc = r + dr, c + dc
                if 0 <= rr < h and 0 <= cc < w and out[rr * w + cc] == 0 and fm[rr * w + cc]:
                    age += 1
                    out[rr * w + cc] = out[idx]
                    heapq.heappush(heap, _Keyed(float(flat[rr * w + cc]), age, rr * w + cc))
        return out.reshape(h, w)

    differs = 0
    for img, mk, mask in _tie_cases()[:8]:
        differs += int(not np.array_equal(skref.watershed(img, mk, mask), watershed_heapq(img, mk, mask)))
    assert differs >= 1


class _Keyed:
    """Heap entry ordered by (value, age) only -- the index never takes part in comparisons."""

    __slots__ = ("value", "age", "index")

    def __init__(self, value, age, index) -> None:
        self.value, self.age, self.index = value, age, index

    def __lt__(self, other) -> bool:
        return (self.value, self.age) < (other.value, other.age)


@pytest.mark.gpu
def test_hip_watershed_forced_ties_match_skimage_heap():
    """``tia_watershed_blobs_f64`` == ``oracle.skref.watershed`` (skimage's ``_watershed_cy`` + ``heap_general.pxi``
    semantics) on inputs with contested (value, age) ties: bit-identical label maps."""
    import torch

    from oracle import skref
    from tiatoolbox_amd.models.architecture import _hover_device as hd

    for k, (img, mk, mask) in enumerate(_tie_cases()):
        exp = skref.watershed(img, mk, mask)
        got = hd.watershed(torch.from_numpy(img).cuda(), torch.from_numpy(mk).cuda(), torch.from_numpy(mask).cuda())
        got = got.cpu().numpy()
        assert got.dtype == np.int32
        assert np.array_equal(got, exp), f"case {k} {img.shape}: {(got != exp).sum()} px differ"
    # batched call: planes of one shape at once, each an independent watershed
    rng = np.random.default_rng(3)
    img = rng.integers(0, 3, (6, 48, 52)).astype(np.float64)
    mk = np.zeros((6, 48, 52), np.int32)
    for i in range(6):
        mk[i].ravel()[rng.choice(48 * 52, 12, replace=False)] = np.arange(1, 13)
    mask = rng.random((6, 48, 52)) < 0.9
    got = hd.watershed(torch.from_numpy(img).cuda(), torch.from_numpy(mk).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
    for i in range(6):
        assert np.array_equal(got[i], skref.watershed(img[i], mk[i], mask[i])), i
    # float data (no ties) and unreachable blobs (no marker inside): stay 0
    img = rng.random((30, 30))
    mk = np.zeros((30, 30), np.int32)
    mk[3, 3] = 7
    mask = np.zeros((30, 30), bool)
    mask[0:10, 0:10] = True
    mask[20:28, 20:28] = True
    got = hd.watershed(torch.from_numpy(img).cuda(), torch.from_numpy(mk).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
    assert np.array_equal(got, skref.watershed(img, mk, mask)) and got[20:28, 20:28].max() == 0


@pytest.mark.gpu
def test_hip_watershed_relaxation_paths_match_the_priority_flood():
    """The three routes of the device watershed against ``oracle.skref.watershed``: blobs of more than 4096 pixels (one
    1024-thread workgroup relaxes them), a long one-pixel-wide serpentine corridor (the relaxation runs out of its pass budget and the
    blob goes to the heap flood), and a quantised plateau field in one large blob (exact ties -> flagged -> heap)."""
    import torch
    from scipy import ndimage

    from oracle import skref
    from tiatoolbox_amd.models.architecture import _hover_device as hd

    def run(img, mk, mask):
        got = hd.watershed(torch.from_numpy(img).cuda(), torch.from_numpy(mk).cuda(), torch.from_numpy(mask).cuda())
        return got.cpu().numpy()

    rng = np.random.default_rng(11)
    # (a) one big blob (~14k px) plus small ones, smooth tie-free values, 25 markers
    h, w = 128, 144
    img = ndimage.gaussian_filter(rng.standard_normal((h, w)), 3.0)
    mask = np.zeros((h, w), bool)
    mask[4:110, 6:140] = True
    mask[60:64, :] = True
    mask[116:126, 10:30] = True
    mask[116:126, 40:45] = True
    seeds = np.zeros((h, w), bool)
    seeds.ravel()[rng.choice(h * w, 40, replace=False)] = True
    mk = ndimage.label(ndimage.binary_dilation(seeds, iterations=1) & mask)[0].astype(np.int32)
    assert (ndimage.label(mask)[0] == ndimage.label(mask)[0][50, 50]).sum() > 4096
    assert np.array_equal(run(img, mk, mask), skref.watershed(img, mk, mask))
    # (b) serpentine corridor, one pixel wide, two markers at its ends: chain length ~ area
    n = 41
    mask = np.zeros((n, n), bool)
    mask[0::2, :] = True
    for r in range(1, n, 2):
        mask[r, n - 1 if (r // 2) % 2 == 0 else 0] = True
    assert ndimage.label(mask)[1] == 1
    img = rng.random((n, n))
    mk = np.zeros((n, n), np.int32)
    mk[0, 0] = 1
    mk[n - 1, n - 1 if ((n - 1) // 2) % 2 == 0 else 0] = 2
    assert np.array_equal(run(img, mk, mask), skref.watershed(img, mk, mask))
    # (c) plateaus: exact ties inside one large blob
    h, w = 96, 96
    img = np.round(ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.0) * 6) / 6
    mask = np.ones((h, w), bool)
    mk = np.zeros((h, w), np.int32)
    mk.ravel()[rng.choice(h * w, 9, replace=False)] = np.arange(1, 10)
    assert np.array_equal(run(img, mk, mask), skref.watershed(img, mk, mask))
    # (d) a batch mixing the cases' plane shapes is covered by the forced-tie test; here: big blob + ties in one batch
    imgs = np.stack([img, ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.0)])
    mks = np.stack([mk, mk])
    masks = np.stack([mask, mask])
    got = run(imgs, mks, masks)
    for i in range(2):
        assert np.array_equal(got[i], skref.watershed(imgs[i], mks[i], masks[i])), i


@pytest.mark.gpu
@pytest.mark.parametrize(("ksize", "scale"), [(21, 1), (11, 0.5)])
def test_hip_stage_planes_vs_oracle(ksize, scale):
    """Intermediate planes of ``_proc_np_hv`` (raw Sobel, distance map, labelled markers, blb) against the oracle's
    locals, stage by stage: a compensating error cannot hide behind an identical final label map."""
    import math

    import torch

    from tiatoolbox_amd.models.architecture import _hover_device as hd

    obj_size = math.ceil(10 * scale**2)
    for (h, w, seed, nb) in ((96, 120, 31, 20), (164, 164, 32, 60)):
        npm, hv, _ = oh.synth_maps(2, h, w, seed=seed, n_blobs=nb)
        got = hd.proc_np_hv_stages(torch.from_numpy(npm).cuda(), torch.from_numpy(hv).cuda(), ksize=ksize, obj_size=obj_size)
        got = {k: v.cpu().numpy() for k, v in got.items()}
        for i in range(2):
            dbg = {}
            exp = oh.proc_np_hv(npm[i], hv[i], scale_factor=scale, debug=dbg)
            assert np.array_equal(got["blb"][i], dbg["blb"])
            assert np.array_equal(got["sobel_h"][i], dbg["sobel_h_raw"]), np.abs(got["sobel_h"][i] - dbg["sobel_h_raw"]).max()
            assert np.array_equal(got["sobel_v"][i], dbg["sobel_v_raw"]), np.abs(got["sobel_v"][i] - dbg["sobel_v_raw"]).max()
            assert np.array_equal(got["dist"][i], dbg["dist"]), np.abs(got["dist"][i] - dbg["dist"]).max()
            assert np.array_equal(got["marker"][i], dbg["marker"])
            assert np.array_equal(got["inst"][i], exp)


@pytest.mark.gpu
def test_hip_instance_info_contours_vs_oracle():
    """get_instance_info (box/centroid/contours/type/prob and the <3-vertex drop rule) on label maps that
    exercise the border follower: noise labels (many components, holes, islands inside holes, pixels on the
    image edge), thin lines, single pixels, nested rings."""
    from tiatoolbox_amd.models.architecture.hovernet import HoVerNet

    rng = np.random.default_rng(3)
    cases = [rng.integers(0, 5, (48, 61)).astype(np.int32), (rng.random((40, 40)) < 0.7).astype(np.int32) * 3,
             np.kron(rng.integers(0, 4, (16, 20)), np.ones((3, 3), np.int64)).astype(np.int32),
             np.kron((rng.random((14, 14)) < 0.6).astype(np.int64), np.ones((2, 3), np.int64)).astype(np.int32)]
    rings = np.zeros((41, 41), np.int32)
    for k, r in enumerate(range(20, 0, -4)):
        rings[20 - r:21 + r, 20 - r:21 + r] = 1 if k % 2 == 0 else 0
    rings[20, 20] = 1
    cases.append(rings)
    shapes = np.zeros((30, 50), np.int32)
    shapes[2, 3:20] = 1       # horizontal line: 2 vertices -> dropped
    shapes[5:25, 45] = 2      # vertical line
    shapes[10, 10] = 3        # single pixel
    for d in range(12):
        shapes[12 + d, 20 + d] = 4   # diagonal line
    shapes[20:28, 2:12] = 5
    shapes[22:26, 4:8] = 0
    shapes[23, 5] = 5         # island inside the hole of the same label
    shapes[0:3, 47:50] = 6    # touches the corner
    cases.append(shapes)
    npm, hv, _ = oh.synth_maps(1, 200, 200, seed=21, n_blobs=80)
    cases.append(oh.proc_np_hv(npm[0], hv[0]))
    for lab in cases:
        pred_type = rng.integers(0, 4, lab.shape).astype(np.uint8)
        for offset in ((0, 0), (100, 7)):
            exp = oh.get_instance_info(lab, pred_type, offset)
            got = HoVerNet.get_instance_info(lab, pred_type, offset)
            assert list(got) == list(exp)
            for k in exp:
                assert np.array_equal(got[k]["box"], exp[k]["box"])
                np.testing.assert_array_equal(got[k]["centroid"], exp[k]["centroid"])
                assert got[k]["contours"].dtype == np.int32
                assert np.array_equal(got[k]["contours"], exp[k]["contours"]), (k, got[k]["contours"], exp[k]["contours"])
                assert got[k]["type"] == exp[k]["type"] and got[k]["prob"] == exp[k]["prob"]
    assert HoVerNet.get_instance_info(np.zeros((8, 8), np.int32)) == {}


@pytest.mark.gpu
def test_hip_proc_np_hv_large_tile_properties():
    """1024x1024 tile (WSI-mode tile size): labels stay inside the mask, every marker id floods a
    connected region, result is deterministic, and a 2x2 mosaic of independent tiles gives the
    per-tile results (blobs never interact across the zero gutter)."""
    import torch
    from scipy import ndimage

    from tiatoolbox_amd.models.architecture import _hover_device as hd

    npm, hv, _ = oh.synth_maps(4, 500, 500, seed=11, n_blobs=400)
    big_np = np.zeros((1, 1024, 1024, 1), np.float32)
    big_hv = np.zeros((1, 1024, 1024, 2), np.float32)
    for k, (y, x) in enumerate(((0, 0), (0, 512), (512, 0), (512, 512))):
        big_np[0, y + 6:y + 506, x + 6:x + 506] = npm[k]
        big_hv[0, y + 6:y + 506, x + 6:x + 506] = hv[k]
    a, _ = hd.proc_np_hv(torch.from_numpy(big_np).cuda(), torch.from_numpy(big_hv).cuda())
    b, _ = hd.proc_np_hv(torch.from_numpy(big_np).cuda(), torch.from_numpy(big_hv).cuda())
    assert torch.equal(a, b)
    lab = a[0].cpu().numpy()
    mask = ndimage.label(big_np[0, ..., 0] >= 0.5)[0]
    assert np.all((lab > 0) <= (mask > 0))
    ids = np.unique(lab)[1:]
    assert len(ids) > 200
    for i in ids[:: max(1, len(ids) // 40)]:
        assert ndimage.label(lab == i)[1] == 1


@pytest.mark.gpu
def test_hip_proc_np_hv_large_tile_vs_oracle():
    """One 1280 x 1280 plane = the WSI-mode tile of ``pretrained_model.yaml:667`` (``tile_shape`` 1024 + 2 x 128 margin) through the
    multi-launch path (plane > 32,000 px: banded Sobel through HBM, watershed by relaxation + heap fall-back) -- the label map must
    EQUAL ``oracle.hovernet.proc_np_hv`` of the same plane, bit for bit (about 3000 nuclei; the nine 426 x 426 synthetic fields are
    laid edge to edge, so blobs also meet across the seams)."""
    import torch

    from tiatoolbox_amd.models.architecture import _hover_device as hd

    npm, hv, _ = oh.synth_maps(9, 426, 426, seed=5, n_blobs=400)
    big_np = np.zeros((1280, 1280, 1), np.float32)
    big_hv = np.zeros((1280, 1280, 2), np.float32)
    for k in range(9):
        y, x = (k // 3) * 426 + 1, (k % 3) * 426 + 1
        big_np[y:y + 426, x:x + 426] = npm[k]
        big_hv[y:y + 426, x:x + 426] = hv[k]
    exp = oh.proc_np_hv(big_np, big_hv)
    assert exp.max() > 2500  # noqa: PLR2004
    got, _ = hd.proc_np_hv(torch.from_numpy(big_np[None]).cuda(), torch.from_numpy(big_hv[None]).cuda())
    got = got[0].cpu().numpy()
    assert got.shape == exp.shape and np.array_equal(got, exp), int((got != exp).sum())


def test_hovernet_state_dict_layout_and_shapes():
    """Parameter names follow the reference (hovernet.py:80-500) so its .pth files load strictly."""
    import torch

    from tiatoolbox_amd.models.architecture.hovernet import DenseBlock, HoVerNet, ResidualBlock, TFSamepaddingLayer

    m = HoVerNet(num_types=6, mode="fast").eval()
    keys = set(m.state_dict())
    for k in ("conv0./.weight", "conv0.bn.running_var", "d0.units.0.conv1.weight", "d0.units.1.preact/bn.weight",
              "d1.units.0.conv2/bn.bias", "d3.shortcut.weight", "conv_bot.weight",
              "decoder.tp.u3.dense.units.0.preact_bna/bn.weight", "decoder.np.u2.convf.weight",
              "decoder.hv.u0.conv.bias", "upsample2x.unpool_mat"):
        assert k in keys, k
    with torch.no_grad():
        out = m(torch.rand(1, 3, 256, 256) * 255)
    assert {k: tuple(v.shape) for k, v in out.items()} == {"tp": (1, 6, 164, 164), "np": (1, 2, 164, 164),
                                                            "hv": (1, 2, 164, 164)}
    # block-level shape checks of the reference's tests/models/test_hovernet.py:65-101
    x = torch.rand(1, 8, 16, 16)
    assert TFSamepaddingLayer(3, 1)(x).shape == (1, 8, 18, 18)
    assert ResidualBlock(8, [1, 3, 1], [16, 16, 32], 2, stride=2)(x).shape == (1, 32, 8, 8)
    assert DenseBlock(8, [1, 3], [16, 16], 3)(x).shape == (1, 8 + 3 * 16, 10, 10)
    with pytest.raises(ValueError, match="Unbalance Unit Info"):
        DenseBlock(8, [1, 3], [16], 3)
    with pytest.raises(ValueError, match="Invalid mode"):
        HoVerNet(mode="xyz")


@pytest.mark.gpu
def test_nucleus_instance_segmentor_patch_mode(conv_algo):
    """Engine end to end on the GPU == CPU model forward + oracle post-processing, patch by patch."""
    import torch

    from tiatoolbox_amd.models.engine.multi_task_segmentor import NucleusInstanceSegmentor
    from tiatoolbox_amd.utils import synth

    patches = synth.g_he(3, 256, 256, seed=13)
    with pytest.warns(DeprecationWarning):
        eng = NucleusInstanceSegmentor("hovernet_fast-pannuke", batch_size=2, device="cuda")
    out = eng.run(patches, patch_mode=True, return_probabilities=True, conv_algo=conv_algo)
    assert set(out) == {"predictions", "box", "centroid", "contours", "prob", "type", "probabilities"}
    assert out["predictions"].shape == (3, 164, 164)
    npm, hv, tp = out["probabilities"]
    assert npm.shape == (3, 164, 164, 1) and hv.shape == (3, 164, 164, 2) and tp.shape == (3, 164, 164, 1)
    cpu_model = eng.model.to("cpu")
    ref_np, ref_hv, ref_tp = cpu_model.infer_batch(cpu_model, torch.from_numpy(patches), device="cpu")
    np.testing.assert_allclose(npm, ref_np, atol=2e-3)
    np.testing.assert_allclose(hv, ref_hv, atol=2e-2, rtol=1e-2)
    for i in range(3):  # post-processing of the GPU heads == oracle on the same heads
        exp = oh.proc_np_hv(npm[i], hv[i])
        assert np.array_equal(out["predictions"][i], exp)
        info = oh.get_instance_info(exp, np.around(tp[i]).astype("uint8")[..., 0])
        assert len(out["box"][i]) == len(info)
        if info:
            assert np.array_equal(out["box"][i], np.array([v["box"] for v in info.values()]))
            np.testing.assert_array_equal(out["centroid"][i], np.array([v["centroid"] for v in info.values()]))
            assert [int(t) for t in out["type"][i]] == [v["type"] for v in info.values()]
    with pytest.raises(ValueError, match="return_labels"):
        eng.run(patches, patch_mode=True, return_labels=True)


def test_columnar_instance_table_equals_per_instance_assembly():
    """Host assembly used by the batched engine path (NumPy columns) == the per-instance restatement of
    hovernet.py:670-748, including type ties, the background runner-up rule and the <3-vertex drop rule."""
    from tiatoolbox_amd.models.architecture import _hover_device as hd

    rng = np.random.default_rng(11)
    m, t = 200, 6
    stats = np.zeros((m + 1, 8), np.int64)
    types = np.zeros((m + 1, t), np.int32)
    meta = np.zeros((m + 1, 4), np.int32)
    chunks, first = [], 0
    for i in range(1, m + 1):
        if rng.random() < 0.1:
            continue  # absent id
        x0, y0 = rng.integers(0, 500, 2)
        bw, bh = rng.integers(1, 40, 2)
        area = int(rng.integers(1, bw * bh + 1))
        stats[i, :7] = [area, x0, y0, x0 + bw - 1, y0 + bh - 1, area * x0 + rng.integers(0, area * bw), area * y0 + rng.integers(0, area * bh)]
        votes = rng.multinomial(area, rng.dirichlet(np.ones(t) * 0.4))
        if rng.random() < 0.3:   # force ties / background wins
            votes[:] = 0
            votes[0] = area // 2
            votes[rng.integers(1, t)] = area - area // 2
        types[i] = votes
        npts = int(rng.integers(1, 12))
        meta[i] = [x0, y0, npts, first]
        chunks.append(rng.integers(0, 600, (npts, 2)).astype(np.int32))
        first += npts
    points = np.concatenate(chunks)
    for offset in ((0, 0), (17, 400)):
        for use_types in (True, False):
            ty = types if use_types else None
            info = hd.info_from_stats(stats, ty, offset, meta=meta, points=points)
            table = hd.table_from_stats(stats, ty, offset, meta=meta, points=points)
            assert list(info) == table["ids"].tolist() and len(info) > 50
            assert np.array_equal(table["box"], np.array([v["box"] for v in info.values()]))
            np.testing.assert_array_equal(table["centroid"], np.array([v["centroid"] for v in info.values()]))
            for j, v in enumerate(info.values()):
                assert np.array_equal(table["contours"][j], v["contours"]) and table["contours"][j].dtype == np.int32
                assert table["type"][j] == v["type"] and table["prob"][j] == v["prob"]
    assert hd.table_from_stats(np.zeros((3, 8), np.int64), None) is None
    # the batch-wide assembly (one NumPy pass per column over all planes) == the per-plane one; an empty plane gives None
    stats_b = np.stack([stats, np.zeros_like(stats), stats[::-1]])
    types_b = np.stack([types, np.zeros_like(types), types[::-1]])
    meta_b = np.stack([meta, np.zeros_like(meta), meta[::-1]])
    for ty in (types_b, None):
        tables = hd.tables_from_stats_batch(stats_b, ty, meta=meta_b, points=points)
        assert len(tables) == 3 and tables[1] is None
        for i in (0, 2):
            one = hd.table_from_stats(stats_b[i], ty[i] if ty is not None else None, meta=meta_b[i], points=points)
            assert np.array_equal(tables[i]["ids"], one["ids"]) and np.array_equal(tables[i]["box"], one["box"])
            np.testing.assert_array_equal(tables[i]["centroid"], one["centroid"])
            for col in ("contours", "type", "prob"):
                assert tables[i][col].dtype == object and len(tables[i][col]) == len(one[col])
                for a, b in zip(tables[i][col], one[col]):
                    assert type(a) is type(b) and np.array_equal(a, b)
    assert hd.tables_from_stats_batch(np.zeros((2, 3, 8), np.int64), None, meta=np.zeros((2, 3, 4), np.int32),
                                      points=np.zeros((0, 2), np.int32)) == [None, None]


def _pop_hole_classic(a: list, smaller) -> tuple:
    """Line-by-line Python mirror of the (skimage-style) pop in ``hover_post.hip`` (hole formulation)."""
    top = a[0]
    items = len(a) - 1
    if items == 0:
        a.pop()
        return top
    last = a[items]
    del a[items]
    at = 0
    while 2 * at + 1 < items:
        left, right = 2 * at + 1, 2 * at + 2
        best, best_i = last, at
        if smaller(a[left], best):
            best, best_i = a[left], left
        if right < items and smaller(a[right], best):
            best, best_i = a[right], right
        if best_i == at:
            break
        a[at] = best
        at = best_i
    a[at] = last
    return top


def test_classic_heap_pop_formulations_agree_on_ties():
    """The kernel's optional skimage-style pop (items shifted through a hole) == the oracle's swap-based restatement
    of ``heap_general.pxi``, pop by pop, on sequences full of (value, age) ties; and the watershed oracle differs from
    a ``heapq``-procedure flood only when such ties exist."""
    import heapq

    from oracle import skref

    rng = np.random.default_rng(2)
    for _ in range(200):
        ref, hole = skref._Heap(), []  # noqa: SLF001
        for step in range(int(rng.integers(5, 120))):
            if hole and rng.random() < 0.4:
                assert ref.pop() == _pop_hole_classic(hole, skref._Heap._smaller)  # noqa: SLF001
            else:
                item = (float(rng.integers(0, 4)), int(rng.integers(0, 3)), step)
                ref.push(item)
                hole.append(item)   # push = append + move up while smaller than the parent (same in both forms)
                c = len(hole) - 1
                while c > 0 and skref._Heap._smaller(hole[c], hole[(c + 1) // 2 - 1]):  # noqa: SLF001
                    p = (c + 1) // 2 - 1
                    hole[c], hole[p] = hole[p], hole[c]
                    c = p
            assert ref.items == hole
        while hole:
            assert ref.pop() == _pop_hole_classic(hole, skref._Heap._smaller)  # noqa: SLF001

    class _Item:  # heapq procedure with the same (value, age) ordering
        def __init__(self, t):
            self.t = t

        def __lt__(self, other):
            return skref._Heap._smaller(self.t, other.t)  # noqa: SLF001

    def flood_heapq(image, markers, mask):
        h, w = image.shape
        out = np.where(mask, markers, 0).astype(np.int32).ravel()
        img, msk, heap, age = image.ravel(), mask.ravel(), [], 0
        for idx in np.flatnonzero(out):
            heapq.heappush(heap, _Item((img[idx], 0, int(idx))))
        while heap:
            idx = heapq.heappop(heap).t[2]
            r, c = divmod(idx, w)
            for dr, dc in ((-1, 0), (0, -1), (0, 1), (1, 0)):
                rr, cc = r + dr, c + dc
                if 0 <= rr < h and 0 <= cc < w and out[rr * w + cc] == 0 and msk[rr * w + cc]:
                    age += 1
                    out[rr * w + cc] = out[idx]
                    heapq.heappush(heap, _Item((img[rr * w + cc], age, rr * w + cc)))
        return out.reshape(h, w)

    img = rng.random((40, 50))
    mask = rng.random((40, 50)) < 0.85
    markers = np.zeros((40, 50), int)
    for k in range(1, 6):
        y, x = rng.integers(0, 36), rng.integers(0, 46)
        markers[y:y + 3, x:x + 3] = k
    assert np.array_equal(skref.watershed(img, markers, mask), flood_heapq(img, markers, mask))  # no ties: identical


@pytest.mark.gpu
@pytest.mark.parametrize("plus", [False, True])
def test_fused_hovernet_forward_matches_plain_module(plus):
    """``FusedHoVerNet`` (104 of 144 convolutions on ``tia_conv2d_nhwc_f32_ex`` with folded BN / fused ReLU / residual
    epilogues, ``tia_scale_shift_act_nhwc_f32`` for the pre-activations, TF "same" padding through the explicit front
    padding) against the plain torch module on the CPU in float32, with randomised BN statistics: every head within
    2e-4 of its range.  Also the two new entry points on their own against torch."""
    import copy

    import torch
    import torch.nn.functional as F  # noqa: N812

    from tiatoolbox_amd.models.architecture.fused import hip_conv2d_ex, hip_scale_shift_act, pack_conv_weights
    from tiatoolbox_amd.models.architecture.hovernet import HoVerNet
    from tiatoolbox_amd.models.architecture.hovernet_fused import FusedHoVerNet
    from tiatoolbox_amd.models.architecture.hovernetplus import HoVerNetPlus

    g = torch.Generator().manual_seed(5)
    if not plus:
        # asymmetric padding: 3x3 stride 2 with 0 rows in front / 1 behind == F.pad + valid convolution; valid 3x3; 1x1 + residual
        for (cin, cout, hw, k, s, lo, hi) in [(64, 128, 20, 3, 2, 0, 1), (32, 64, 17, 3, 1, 0, 0), (96, 64, 9, 1, 1, 0, 0),
                                              (64, 64, 12, 3, 1, 1, 1), (32, 64, 11, 3, 2, 1, 1)]:
            conv = torch.nn.Conv2d(cin, cout, k, stride=s, bias=True)
            x = torch.randn((3, cin, hw, hw), generator=g)
            ref = conv(F.pad(x, (lo, hi, lo, hi))).detach()
            res = torch.randn(ref.shape, generator=g)
            dev = copy.deepcopy(conv).cuda()
            got = hip_conv2d_ex(x.cuda().contiguous(memory_format=torch.channels_last), pack_conv_weights(dev), dev.bias,
                                res.cuda().contiguous(memory_format=torch.channels_last), kernel=k, stride=s, pad_lo=lo, pad_hi=hi,
                                relu=True)
            assert (got.cpu() - torch.relu(ref + res)).abs().max() <= 1e-4, (cin, cout, hw, k, s, lo, hi)
        from tiatoolbox_amd.models.architecture.fused import hip_conv2d_post

        for cout in (64, 256):  # second output of the epilogue: relu(bn(conv + residual)), with and without the raw sum
            conv = torch.nn.Conv2d(96, cout, 1, bias=False)
            x = torch.randn((2, 96, 9, 6), generator=g)
            res = torch.randn((2, cout, 9, 6), generator=g)
            ps, pt = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
            raw_ref = conv(x).detach() + res
            act_ref = torch.relu(raw_ref * ps[None, :, None, None] + pt[None, :, None, None])
            dev = copy.deepcopy(conv).cuda()
            for want_raw in (True, False):
                raw, act = hip_conv2d_post(x.cuda().contiguous(memory_format=torch.channels_last), pack_conv_weights(dev), None,
                                           res.cuda().contiguous(memory_format=torch.channels_last), kernel=1, stride=1, pad_lo=0,
                                           pad_hi=0, relu=False, post_scale=ps.cuda(), post_shift=pt.cuda(), want_raw=want_raw)
                assert (raw is None) == (not want_raw)
                assert (act.cpu() - act_ref).abs().max() <= 1e-4
                if raw is not None:
                    assert (raw.cpu() - raw_ref).abs().max() <= 1e-4
                    again = torch.relu(raw * ps.cuda()[None, :, None, None] + pt.cuda()[None, :, None, None])
                    assert (act - again).abs().max() <= 1e-6  # the two outputs are consistent with each other
        from tiatoolbox_amd.models.architecture.fused import hip_conv1x1_pre

        # activation on load (conv1 of residual units 2..n): == scale_shift_act followed by the plain convolution, bit for bit
        # in the operand (same separately rounded product and sum), with and without stride / residual, both tile widths
        for (cin, cout, s, with_res) in [(96, 64, 1, False), (256, 128, 1, True), (512, 256, 2, False), (1024, 512, 1, False)]:
            conv = torch.nn.Conv2d(cin, cout, 1, stride=s, bias=True)
            x = torch.randn((3, cin, 11, 7), generator=g)
            sc, sh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
            ref = conv(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])).detach()
            res = torch.randn(ref.shape, generator=g) if with_res else None
            dev = copy.deepcopy(conv).cuda()
            xd = x.cuda().contiguous(memory_format=torch.channels_last)
            rd = res.cuda().contiguous(memory_format=torch.channels_last) if with_res else None
            got = hip_conv1x1_pre(xd, sc.cuda(), sh.cuda(), pack_conv_weights(dev), dev.bias, rd, stride=s, relu=True)
            want = torch.relu(ref + res) if with_res else torch.relu(ref)
            assert got.shape == want.shape
            assert (got.cpu() - want).abs().max() <= 2e-4 * max(1.0, float(want.abs().max())), (cin, cout, s)
            two_step = hip_conv2d_ex(hip_scale_shift_act(xd, sc.cuda(), sh.cuda()), pack_conv_weights(dev), dev.bias, rd, kernel=1,
                                     stride=s, pad_lo=0, pad_hi=0, relu=True)
            assert (got - two_step).abs().max() <= 1e-5 * max(1.0, float(want.abs().max())), (cin, cout, s)
        x = torch.randn((2, 96, 7, 5), generator=g)
        sc, sh = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g)
        got = hip_scale_shift_act(x.cuda().contiguous(memory_format=torch.channels_last), sc.cuda(), sh.cuda())
        assert torch.equal(got.cpu(), torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]))
        from tiatoolbox_amd.models.architecture.fused import hip_scale_shift_act_view, hip_upsample2x_add

        from tiatoolbox_amd.models.architecture.fused import hip_grouped_conv_valid

        for k in (3, 5):  # the dense units' grouped convolution (fast: 3x3, original: 5x5), also into a slice of a wider buffer
            conv = torch.nn.Conv2d(128, 32, k, groups=4, bias=False)
            xg = torch.randn((3, 128, 13, 17), generator=g)
            ref = conv(xg).detach()
            wp = conv.weight.detach().view(4, 8, 32, k, k).permute(0, 3, 4, 2, 1).contiguous().cuda()
            xd = xg.cuda().contiguous(memory_format=torch.channels_last)
            got = hip_grouped_conv_valid(xd, wp, groups=4, kernel=k)
            assert got.shape == ref.shape and (got.cpu() - ref).abs().max() <= 1e-5
            big = torch.zeros((3, 96, 13, 17), device="cuda").contiguous(memory_format=torch.channels_last)
            r = (k - 1) // 2
            hip_grouped_conv_valid(xd, wp, groups=4, kernel=k, out=big[:, 64:96, r:13 - r, r:17 - r])
            assert torch.equal(big[:, 64:96, r:13 - r, r:17 - r], got) and float(big[:, :64].abs().max()) == 0.0
        wide = torch.randn((2, 160, 9, 11), generator=g).cuda().contiguous(memory_format=torch.channels_last)
        win = wide[:, :96, 2:7, 1:10]
        exp = torch.relu(win * sc.cuda()[None, :, None, None] + sh.cuda()[None, :, None, None])
        got = hip_scale_shift_act_view(win, sc.cuda(), sh.cuda())
        assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, exp)

        lo = torch.randn((2, 32, 5, 7), generator=g).cuda().contiguous(memory_format=torch.channels_last)
        skip = torch.randn((2, 32, 16, 20), generator=g).cuda().contiguous(memory_format=torch.channels_last)[:, :, 3:13, 3:17]
        assert not skip.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(hip_upsample2x_add(lo, skip), lo.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + skip)
        s32, t32 = (torch.rand(32, generator=g) + 0.5).cuda(), torch.randn(32, generator=g).cuda()
        both = torch.relu((lo.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + skip) * s32[None, :, None, None]
                          + t32[None, :, None, None])
        assert (hip_upsample2x_add(lo, skip, s32, t32) - both).abs().max() <= 1e-6

    torch.manual_seed(3)
    model = (HoVerNetPlus(num_types=3, num_layers=5) if plus else HoVerNet(num_types=6, mode="fast")).eval()
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.05, generator=g)
            mod.running_var.uniform_(0.8, 1.2, generator=g)
            mod.weight.data.uniform_(0.8, 1.2, generator=g)
            mod.bias.data.normal_(0, 0.05, generator=g)
    x = torch.randint(0, 256, (1 if plus else 2, 3, 256, 256), generator=g).float()
    with torch.inference_mode():
        ref = model(x)
        fused = FusedHoVerNet(copy.deepcopy(model).cuda()).cuda()
        got = fused(x.cuda().contiguous(memory_format=torch.channels_last))
    assert list(got) == list(ref)
    for name in ref:
        r, o = ref[name], got[name].cpu()
        assert o.shape == r.shape == (x.shape[0], r.shape[1], 164, 164)
        assert (o - r).abs().max() <= 2e-4 * max(float(r.abs().max()), 1.0), name
