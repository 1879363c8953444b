"""bench.py's `cpu_baseline` leg (the CPU oracle on a bounded sample, thread count calibrated) on a toy configuration: the JSON
object the driver's bench line carries must have its fields whatever the host looks like."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_leg_returns_the_contract_fields():
    import bench

    before = torch.get_num_threads()
    a = argparse.Namespace(hidden=32, layers=4, types=4, dropout=0.1, nodes=60, messages=240)
    out = bench.cpu_baseline(a, seconds_budget=5.0)
    assert torch.get_num_threads() == before  # the calibration restores the thread count
    assert out["kind"] == "port" and out["unit"] == "graphs/s" and out["value"] > 0
    assert 1 <= out["cores"] <= (os.cpu_count() or 1) and out["host_threads"] == (os.cpu_count() or 8)
    assert "4-graph minibatch" in out["sample"] and out["collate_ms"] >= 0
