"""Seeded random shapes through the tap-reuse kernel's geometries (16 x 16 blocks, two 8 x 8 images, bands of virtual / real rows,
"same" and valid borders, rectangular maps, ragged batches) against the slice kernel of the same library: the two-output epilogue form
of ``tia_conv2d_nhwc_f32`` always runs on the register-staged slice kernel, whose raw output is the same convolution with the same
float32 accumulation over (tap, 16-channel slice) -- what differs is the channel order inside a slice, i.e. rounding noise.  The band
geometry's bookkeeping (strip / band / image-boundary arithmetic with host-computed reciprocals) is what this guards; the fixed cases
against ``torch.nn.functional.conv2d`` on the CPU live in ``tests/test_engine.py``."""
import ctypes

import numpy as np
import pytest
import torch


@pytest.mark.gpu
def test_tap_reuse_geometries_equal_slice_kernel_on_random_shapes():
    from tiatoolbox_amd import _lib
    from tiatoolbox_amd.models.architecture.fused import hip_conv2d_ex, hip_conv2d_post

    rng = np.random.default_rng(20260926)
    g = torch.Generator(device="cuda").manual_seed(5)
    kinds = {}
    for case in range(72):
        pad = int(rng.integers(0, 2))
        cin, cout = int(rng.choice([32, 64, 96])), int(rng.choice([64, 128, 192]))
        if case % 4 == 0:    # maps whose width factors into strips: bands
            w = int(rng.choice([7, 14, 21, 28, 30, 33, 42, 45, 56, 60, 90])) + (0 if pad else 2)
            h = int(rng.integers(4, 70))
        elif case % 4 == 1:  # anything
            h, w = int(rng.integers(4, 80)), int(rng.integers(4, 80))
        elif case % 4 == 2:  # square maps of the networks' odd sizes
            h = w = int(rng.choice([5, 7, 9, 14, 17, 28, 34, 41, 56, 68, 82, 120])) + (0 if pad else 2)
        else:                # block-friendly maps: 16 x 16 blocks, two 8 x 8 images
            h = w = int(rng.choice([8, 16, 32, 48, 64])) + (0 if pad else 2)
        n = int(rng.integers(1, 12))
        ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
        geom = (ctypes.c_int32 * 4)()
        kind = _lib.load().tia_conv3x3_geometry(h, w, ho, wo, pad, pad, geom)
        kinds[kind] = kinds.get(kind, 0) + 1
        x = torch.randn((n, cin, h, w), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        wt = torch.randn((3, 3, cin, cout), device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
        bias = torch.randn(cout, device="cuda", generator=g) * 0.1
        res = torch.randn((n, cout, ho, wo), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        got = hip_conv2d_ex(x, wt, bias, res, kernel=3, stride=1, pad_lo=pad, pad_hi=pad, relu=False)
        raw, _ = hip_conv2d_post(x, wt, bias, res, kernel=3, stride=1, pad_lo=pad, pad_hi=pad, relu=False,
                                 post_scale=torch.ones(cout, device="cuda"), post_shift=torch.zeros(cout, device="cuda"))
        assert got.shape == raw.shape == (n, cout, ho, wo)
        err = (got - raw).abs().max().item()
        assert err <= 3e-5, (case, kind, list(geom), n, cin, cout, h, w, pad, err)  # noqa: PLR2004
    # the sweep must have exercised the band geometry of real rows, the fixed geometries and the slice-kernel fall-through
    assert kinds.get(4, 0) >= 10 and kinds.get(1, 0) >= 5 and kinds.get(2, 0) >= 1 and kinds.get(0, 0) >= 1, kinds  # noqa: PLR2004
