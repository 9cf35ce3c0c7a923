"""Offline known answers of small helpers on the path (SURVEY 8(c) pins): ``cast_to_min_dtype`` exactly as the reference's
own test parametrises it (``/root/reference/tests/test_utils.py:2254-2295``), on NumPy arrays and torch tensors; the
constant-shape row batching of the WSI loops."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tiatoolbox_amd.utils.misc import cast_to_min_dtype

CASES = [
    (np.array([0, 1]), np.bool_),  # Should cast to bool
    (np.array([0, 255]), np.uint8),  # Should cast to uint8
    (np.array([0, 256]), np.uint16),  # Should cast to uint16
    (np.array([0, 70000]), np.uint32),  # Should cast to uint32
    (np.array([0, 2**32]), np.uint64),  # Should cast to uint64
]


@pytest.mark.parametrize(("input_array", "expected_dtype"), CASES)
def test_cast_to_min_dtype_numpy(input_array, expected_dtype):
    """reference tests/test_utils.py:2254-2268"""
    result = cast_to_min_dtype(input_array)
    assert isinstance(result, np.ndarray)
    assert result.dtype == expected_dtype
    assert np.array_equal(result.astype(np.uint64), input_array.astype(np.uint64))


@pytest.mark.parametrize(("input_array", "expected_dtype"), CASES)
def test_cast_to_min_dtype_torch(input_array, expected_dtype):
    """The engines call it on device-resident ``argmax`` results (the reference: dask arrays, :2271-2285): same dtypes."""
    t = torch.from_numpy(input_array)
    result = cast_to_min_dtype(t)
    assert isinstance(result, torch.Tensor)
    expected = {np.bool_: torch.bool, np.uint8: torch.uint8, np.uint16: torch.uint16, np.uint32: torch.uint32,
                np.uint64: torch.uint64}[expected_dtype]
    assert result.dtype == expected
    assert np.array_equal(result.cpu().numpy().astype(np.uint64), input_array.astype(np.uint64))


def test_cast_to_min_dtype_numpy_large_value():
    """reference tests/test_utils.py:2288-2293: beyond uint64 the array comes back unchanged."""
    large_value = np.array([np.iinfo(np.uint64).max + 1], dtype=object)
    result = cast_to_min_dtype(large_value)
    assert result == large_value
    assert result.dtype == object


def test_iter_row_outputs_constant_batch_shape():
    """``iter_row_outputs``: every inference call sees exactly ``batch_size`` patches (the last one padded by repeating its
    final patch), rows come back complete and in order, empty rows as ``None``, single tensors and head tuples alike."""
    from tiatoolbox_amd.models.engine.engine_abc import iter_row_outputs

    rng = np.random.default_rng(0)
    for batch_size in (1, 3, 8):
        for tuple_out in (False, True):
            lens = [5, 0, 9, 1, 0, 0, 7, 2]
            ids = rng.permutation(sum(lens) + 10)[: sum(lens)]
            row_sels, p = [], 0
            for n in lens:
                row_sels.append(ids[p:p + n])
                p += n
            calls = []

            def infer(idx, tuple_out=tuple_out, calls=calls, batch_size=batch_size):
                assert len(idx) == batch_size
                calls.append(np.array(idx))
                a = torch.as_tensor(idx, dtype=torch.float32).view(-1, 1) * torch.ones(1, 4)
                return (a, -a[:, :2]) if tuple_out else a

            got = list(iter_row_outputs(infer, row_sels, batch_size))
            assert [k for k, _ in got] == list(range(len(lens)))
            for (k, out), sel in zip(got, row_sels):
                if len(sel) == 0:
                    assert out is None
                    continue
                first = out[0] if tuple_out else out
                assert first.shape == (len(sel), 4) and np.array_equal(first[:, 0].numpy(), sel.astype(np.float32))
                if tuple_out:
                    assert np.array_equal(out[1][:, 1].numpy(), -sel.astype(np.float32))
            flat = np.concatenate(calls)
            total = sum(lens)
            assert len(calls) == -(-total // batch_size) and np.array_equal(flat[:total], ids)
            assert (flat[total:] == ids[-1]).all()
    assert list(iter_row_outputs(lambda idx: None, [np.zeros(0, int)] * 3, 4)) == [(0, None), (1, None), (2, None)]
