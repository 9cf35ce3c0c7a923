"""Host-side helpers of ``bench.py`` that turn committed rocprofv3 counter summaries into the ``roofline.traffic`` field."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def test_pmc_traffic_per_dispatch_and_per_layer_call(tmp_path, monkeypatch):
    """``bench.pmc_traffic``: FETCH_SIZE counts half the bytes on gfx950 (x 2), WRITE_SIZE as is, units KiB; the newest pass
    (last in name order) of a stem is taken.  Without a ``# PMC forwards=N`` line the figure is the mean per DISPATCH of the
    kernels whose name contains the substring; with it (and the kernel's calls per forward) it is the counter total per layer
    CALL -- a 4096-patch call is several dispatches of < 2 GiB input each."""
    import bench

    prof = tmp_path / "profiles"
    prof.mkdir()
    body = ("# x_counter_collection.csv  (rocprofv3 --pmc), mean per dispatch\n"
            "    {c} mean=          1000.0 n=   30  void (anonymous namespace)::conv3x3_spatial_kernel<64, 0, G16>(void const*)\n"
            "    {c} mean=          3000.0 n=    9  void (anonymous namespace)::conv3x3_spatial_kernel<128, 0, G8>(void const*)\n"
            "    {c} mean=         99999.0 n=    5  void (anonymous namespace)::conv1x1_ring_kernel<128>(float const*)\n")
    for tag, marker in (("r09a", ""), ("r09b", "# PMC forwards=3\n")):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            (prof / f"{tag}_trunkX_pmc_{c}.txt").write_text(body.format(c=c) + marker)
    (prof / "r09b_COMMIT.txt").write_text("abc1234\n")
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    total_kib = 30 * 1000.0 + 9 * 3000.0
    got = bench.pmc_traffic("conv3x3_spatial_kernel", "trunkX", 13)  # newest pass: per layer call
    assert got["bytes"] == pytest.approx((2.0 + 1.0) * total_kib / (3 * 13) * 1024.0)
    assert "r09b_trunkX_pmc_FETCH_SIZE.txt" in got["source"] and "per layer call" in got["source"] and "abc1234" in got["source"]
    per_dispatch = bench.pmc_traffic("conv3x3_spatial_kernel", "trunkX")  # the caller does not know the calls per forward
    assert per_dispatch["bytes"] == pytest.approx(3.0 * total_kib / 39 * 1024.0) and "mean per dispatch" in per_dispatch["source"]
    (prof / "r09b_trunkX_pmc_FETCH_SIZE.txt").unlink()
    (prof / "r09b_trunkX_pmc_WRITE_SIZE.txt").unlink()
    older = bench.pmc_traffic("conv3x3_spatial_kernel", "trunkX", 13)  # a pass without the marker: mean per dispatch
    assert older["bytes"] == pytest.approx(3.0 * total_kib / 39 * 1024.0) and "r09a_" in older["source"]
    assert bench.pmc_traffic("no_such_kernel", "trunkX") is None and bench.pmc_traffic("conv3x3_spatial_kernel", "absent") is None
