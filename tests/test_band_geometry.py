"""The band geometry of the tap-reuse convolution kernel (``conv3x3_spatial.hip``, plan kinds 3 / 4), replayed on the CPU.

The library answers ``tia_conv3x3_geometry`` on the host (strip width ``bw``, rows per band ``br``, LDS row pitch, strips); this test
walks the blocks of a launch exactly as the kernel does -- band -> first real row ``R0``, virtual rows ``v(rr) = rr + gap (rr / ho)``
with image pitch ``h + pad``, the LDS patch = virtual rows ``v(R0) - pad .. v(last) - pad + 2`` x columns ``tx0 - pad .. tx0 + bw + 1 -
pad`` with everything outside an image read as zero, GEMM row ``m`` -> (patch row ``v(R0 + m / bw) - v(R0)``, column ``m % bw``), epilogue
row -> (image, oy, ox) -- and checks that (i) the patch fits the kernel's 1,728-unit buffer, (ii) every output pixel of the batch is
produced exactly once, (iii) the values equal ``torch.nn.functional.conv2d``.  It guards the index arithmetic (and the plan's patch
budget) without a GPU; the kernel itself is compared with torch and with the slice kernel in the ``-m gpu`` tests."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F  # noqa: N812


def _plan(h, w, pad):
    from tiatoolbox_amd import _lib, build

    if not build.LIB_PATH.exists():
        pytest.skip("library not built (run __graft_entry__.build())")
    geom = (ctypes.c_int32 * 4)()
    kind = _lib.load().tia_conv3x3_geometry(h, w, h + 2 * pad - 2, w + 2 * pad - 2, pad, pad, geom)
    return kind, list(geom)


def _replay(x: np.ndarray, wt: np.ndarray, pad: int, kind: int, bw: int, br: int, pitch: int, strips: int) -> np.ndarray:
    n, h, w, cin = x.shape
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    cout = wt.shape[-1]
    assert bw * strips == wo and bw * br <= 256 and pitch >= 4 * (bw + 2)  # noqa: PLR2004
    image_pitch = h + pad                 # virtual rows per image ("same": the shared zero row; valid: none)
    gap = image_pitch - ho
    packed = kind == 4                    # noqa: PLR2004
    total_rows = n * ho if packed else n * (h + 1) - 1
    y = np.zeros((n, ho, wo, cout), dtype=np.float64)
    seen = np.zeros((n, ho, wo), dtype=np.int32)
    v = lambda rr: rr + gap * (rr // ho)  # noqa: E731
    for band in range((total_rows + br - 1) // br):
        for strip in range(strips):
            tx0 = strip * bw
            if packed:
                r0 = band * br
                ty0 = v(r0)
                vlast = v(min(r0 + br, n * ho) - 1) - pad + 2
            else:
                r0, ty0 = 0, band * br
                vlast = ty0 + br
            rows = vlast - (ty0 - pad) + 1
            assert rows * pitch <= 1728, (rows, pitch)  # noqa: PLR2004  the kernel's patch buffer (16-byte units)
            patch = np.zeros((rows, bw + 2, cin), dtype=np.float64)
            for py in range(rows):
                vy = ty0 - pad + py
                g, iy = divmod(vy, image_pitch) if vy >= 0 else (0, -1)
                if vy < 0 or g >= n or iy >= h:
                    continue  # before the first image, behind the last one, or the zero row between two images
                for px in range(bw + 2):
                    ix = tx0 - pad + px
                    if 0 <= ix < w:
                        patch[py, px] = x[g, iy, ix]
            for m in range(br * bw):
                r, col = divmod(m, bw)
                if packed:
                    rr = r0 + r
                    if rr >= n * ho:
                        continue  # (the kernel computes pixel 0 again and drops it)
                    prow, g, oy = v(rr) - ty0, rr // ho, rr % ho
                else:
                    vy = ty0 + r
                    g, oy = divmod(vy, image_pitch)
                    prow = r
                    if oy >= ho or g >= n:
                        continue  # the zero row between two images: computed and dropped
                acc = np.zeros(cout)
                for ky in range(3):
                    for kx in range(3):
                        acc += patch[prow + ky, col + kx] @ wt[ky, kx]
                y[g, oy, tx0 + col] = acc
                seen[g, oy, tx0 + col] += 1
    assert (seen == 1).all()
    return y


@pytest.mark.parametrize(("n", "h", "w", "pad"), [
    (9, 7, 7, 1),      # 7-wide bands of 36 real rows: five image boundaries inside one patch
    (5, 14, 14, 1), (3, 28, 28, 1), (2, 56, 56, 1),
    (3, 21, 42, 1),    # rectangular
    (4, 9, 17, 0),     # valid: two input rows between neighbouring images
    (2, 32, 92, 0),    # HoVer-Net's decoder width (90 = 6 strips of 15)
    (7, 5, 5, 1), (1, 120, 120, 1),
])
def test_band_blocks_cover_the_batch_once_and_equal_conv2d(n, h, w, pad):
    kind, (bw, br, pitch, strips) = _plan(h, w, pad)
    if kind not in (3, 4):
        pytest.skip(f"{h}x{w} pad {pad}: fixed geometry / slice kernel (kind {kind})")
    rng = np.random.default_rng(h * 1000 + w)
    cin, cout = 3, 4
    x = rng.standard_normal((n, h, w, cin))
    wt = rng.standard_normal((3, 3, cin, cout))
    got = _replay(x, wt, pad, kind, bw, br, pitch, strips)
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(wt).permute(3, 2, 0, 1), padding=pad)
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), rtol=0, atol=1e-12)
