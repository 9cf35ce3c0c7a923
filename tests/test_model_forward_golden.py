"""``UNetModel`` against the reference's own module (``/root/reference/tiatoolbox/models/architecture/unet.py:243-476``).

``tests/golden/model_forward_golden.npz`` was produced by ``tests/golden/make_golden.py models``: the REFERENCE's ``UNetModel``
(decoder, skip connections, up-sampling, ``infer_batch`` are its own code; torchvision's ``ResNet`` base class, absent in the build
container, is bound to a restatement by ``_refshim.bind_torchvision_resnet``) loaded strictly with this repo's seeded parameters.
This repo's plain module must reproduce those outputs -- it is what the fused GPU inference copy (``FusedUNet``) is tested
against (``tests/test_semantic.py``)."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

from tiatoolbox_amd.utils import synth

GOLD = Path(__file__).parent / "golden" / "model_forward_golden.npz"


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def _input():
    return torch.from_numpy(synth.g_he(1, 256, 256, seed=78)).float().permute(0, 3, 1, 2)


def test_unet_resnet50_forward_and_infer_batch_match_the_reference_module(gold):
    from tiatoolbox_amd.models.architecture.unet import UNetModel

    x = _input()
    torch.manual_seed(7)
    m = UNetModel(3, 5, "resnet50", decoder_block=[3, 3]).eval()
    with torch.no_grad():
        o = m(x)
    assert tuple(o.shape) == tuple(int(v) for v in gold["fwd_unet_shape"])
    np.testing.assert_allclose(o[0, :, ::8, ::8].numpy(), gold["fwd_unet"], atol=1e-5, rtol=1e-5)
    probs = UNetModel.infer_batch(m, x.permute(0, 2, 3, 1).numpy(), device="cpu")
    probs = np.asarray(probs[0] if isinstance(probs, (list, tuple)) else probs)
    assert tuple(probs.shape) == tuple(int(v) for v in gold["fwd_unet_infer_shape"])
    np.testing.assert_allclose(probs[0, ::8, ::8], gold["fwd_unet_infer"], atol=1e-6, rtol=1e-5)


def test_unet_plain_encoder_forward_matches_the_reference_module(gold):
    from tiatoolbox_amd.models.architecture.unet import UNetModel

    torch.manual_seed(9)
    m = UNetModel(3, 2, "unet", decoder_block=[3]).eval()
    with torch.no_grad():
        o = m(_input())
    np.testing.assert_allclose(o[0, :, ::8, ::8].numpy(), gold["fwd_unet_plain"], atol=1e-5, rtol=1e-5)


@pytest.mark.gpu
def test_fused_unet_reproduces_the_reference_module_output(gold):
    """The GPU inference copy (stem kernel, MFMA convolutions, head kernel) against the reference module's output directly."""
    from tiatoolbox_amd.models.architecture.unet import UNetModel
    from tiatoolbox_amd.models.architecture.unet_fused import FusedUNet

    torch.manual_seed(7)
    m = UNetModel(3, 5, "resnet50", decoder_block=[3, 3]).eval()
    with torch.inference_mode():
        got = FusedUNet(m.cuda()).cuda()(_input().cuda().contiguous(memory_format=torch.channels_last)).cpu()
    ref = gold["fwd_unet"]
    err = np.abs(got[0, :, ::8, ::8].numpy() - ref).max()
    assert err <= 2e-4 * max(1.0, float(np.abs(ref).max())), err
