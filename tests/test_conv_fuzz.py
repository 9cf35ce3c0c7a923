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


@pytest.mark.gpu
def test_batches_beyond_2gib_of_input_run_in_equal_groups_bit_identically():
    """The convolution kernels address their input with 32-bit byte offsets, so the entry points cut a batch into equal groups of
    < 2 GiB of input (``tia::even_group``; the bench's 4096-patch engine batch does this on every 64-channel layer).  Each entry
    point -- float32 tap reuse, Winograd, the half kernels, the stem with float32 input -- over a batch of ~2.2 GiB must equal, bit for
    bit, the same call over three sub-batches that need no split (shapes whose kernel choice does not depend on the batch size)."""
    import gc

    from tiatoolbox_amd.models.architecture.fused import (hip_conv2d, hip_conv2d_h, hip_conv3x3_wino, hip_stem_conv_pool,
                                                          pack_conv_weights, pack_conv_weights_h, pack_conv_weights_wino,
                                                          pack_stem_weights)

    g = torch.Generator(device="cuda").manual_seed(11)
    n = 2100

    def thirds(fn, x, res=None):
        cuts = (0, 700, 1400, n)
        return torch.cat([fn(x[a:b], None if res is None else res[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])

    conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn((n, 64, 64, 64), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)  # 1 MiB per image
    res = torch.randn((n, 64, 64, 64), device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    assert x.numel() * 4 > 2**31
    wp, up = pack_conv_weights(conv), pack_conv_weights_wino(conv)
    direct = lambda xs, rs: hip_conv2d(xs, wp, conv.bias, rs, kernel=3, stride=1, padding=1, relu=True)  # noqa: E731
    wino = lambda xs, rs: hip_conv3x3_wino(xs, up, conv.bias, rs, padding=1, relu=True)  # noqa: E731
    assert torch.equal(direct(x, res), thirds(direct, x, res))
    assert torch.equal(wino(x, res), thirds(wino, x, res))
    del x, res
    gc.collect()
    torch.cuda.empty_cache()

    conv_h = torch.nn.Conv2d(128, 128, 3, padding=1).cuda()
    xh = torch.randn((n, 128, 64, 64), device="cuda", generator=g).to(torch.float16).contiguous(memory_format=torch.channels_last)
    assert xh.numel() * 2 > 2**31
    wh = pack_conv_weights_h(conv_h, torch.float16)
    half = lambda xs, rs: hip_conv2d_h(xs, wh, conv_h.bias.detach(), rs, cout=128, kernel=3, stride=1, padding=1, relu=True)  # noqa: E731
    assert torch.equal(half(xh, None), thirds(half, xh))
    del xh
    gc.collect()
    torch.cuda.empty_cache()

    stem = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3).cuda()
    ns = 2800
    xs = torch.rand((ns, 256, 256, 3), device="cuda", generator=g)  # float32 NHWC: 768 KiB per image
    assert xs.numel() * 4 > 2**31
    ws = pack_stem_weights(stem)
    full = hip_stem_conv_pool(xs, ws, stem.bias.detach())
    parts = torch.cat([hip_stem_conv_pool(xs[a:b].contiguous(), ws, stem.bias.detach()) for a, b in ((0, 900), (900, 1800), (1800, ns))])
    assert torch.equal(full, parts)
