"""Host-side pieces of bench.py that need no GPU: which committed profile a line quotes (newest ROUND, not the
lexicographically largest name -- r100 > r99, r03b > r03), the stale-kernel flag, the PMC traffic lookup."""
import json
import os

import bench


def test_measured_parity_picks_the_newest_round_and_flags_stale_kernels(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    body = lambda tag, stamp: dict(_meta=dict(kernel_source_hash=stamp), summary={"bf16x3": {"metric_depth": {"tag": tag}}})
    from sparf_amd.build import source_hash
    for name, tag, stamp in (("r99_parity_scale.json", "r99", "0" * 16), ("r100_parity_scale.json", "r100", source_hash()),
                             ("r03_parity_scale.json", "r03", None), ("r03b_parity_scale.json", "r03b", "0" * 16)):
        d = body(tag, stamp)
        if stamp is None:
            d["_meta"].pop("kernel_source_hash")
        (prof / name).write_text(json.dumps(d))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.measured_parity("bf16x3")
    assert got["metric_depth"]["tag"] == "r100" and got["stale"] is False and got["source"] == os.path.join("profiles", "r100_parity_scale.json")
    (prof / "r100_parity_scale.json").unlink()
    got = bench.measured_parity("bf16x3")
    assert got["metric_depth"]["tag"] == "r99" and got["stale"] is True          # measured on other kernel sources
    (prof / "r99_parity_scale.json").unlink()
    assert bench.measured_parity("bf16x3")["metric_depth"]["tag"] == "r03b"        # suffix sessions of a round sort after the plain tag
    (prof / "r03b_parity_scale.json").unlink()
    got = bench.measured_parity("bf16x3")
    assert got["metric_depth"]["tag"] == "r03" and got["stale"] is None           # unstamped profile: unknown
    assert bench.measured_parity("fp32") is None


def test_pmc_traffic_reads_the_committed_profile():
    total, src, util = bench.pmc_traffic("mlp_fwd", "bf16x3", 786432)
    assert src is not None and src.startswith("profiles/") and 3.5e9 < total < 4.5e9
    # matrix-pipe busy: ~62 % of the cycles at the ~1.85-1.9 GHz the chip held = ~48 % of the issue slots at the 2.4 GHz peak clock
    assert 0.3 < util["at_clock_held"] < 0.8 and util["at_peak_clock"] < util["at_clock_held"] and 1.5 < util["clock_held_ghz"] < 2.45
    assert bench.pmc_traffic("mlp_fwd", "bf16x3", 12345) == (None, None, None)      # no profile at that row count
