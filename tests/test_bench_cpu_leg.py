"""bench.py's `cpu_baseline` leg (the CPU oracle on a bounded sample, thread count calibrated) on a toy configuration: the JSON
object the driver's bench line carries must have its fields whatever the host looks like."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_leg_returns_the_contract_fields():
    """the CPU oracle through ModelTrainer.train on a synthetic shard (oracle/cpu_train.py), here on a toy
    configuration -- the record must carry the contract's fields, say what the sample was, and leave the thread count alone."""
    import bench

    before = torch.get_num_threads()
    a = argparse.Namespace(hidden=32, layers=4, types=4, dropout=0.1, nodes=120, messages=0, placement="message")
    out = bench.cpu_baseline(a, seconds_budget=4.0, graphs_per_minibatch=3)
    assert torch.get_num_threads() == before
    assert out["kind"] == "port" and out["unit"] == "graphs/s" and out["value"] > 0
    assert 1 <= out["cores"] <= (os.cpu_count() or 1) and out["host_threads"] == (os.cpu_count() or 8)
    assert out["minibatch"]["graphs"] == 3 and out["minibatch"]["edge_types"] > 4 and out["epochs_timed"] >= 1
    assert "ModelTrainer.train" in out["sample"] and "placement message" in out["sample"] and out["input_ms_per_minibatch"] >= 0
    assert out["end_to_end_graphs_per_s"] <= out["value"] * 1.001


def test_pmc_record_of_a_kind_is_the_launch_weighted_mean_over_its_kernels(tmp_path):
    """The message GEMM runs as two kernels (128 x 128 tile for the hidden-128 layers, the wide one for >= 256 output columns):
    bench.py's `traffic` / `mfma_busy_frac` for the kind are launch-weighted means over both records of the PMC summary."""
    import json

    import bench

    f = tmp_path / "r99_pmc.json"
    f.write_text(json.dumps({"kernels": {
        "void gemm_rows_x6_kernel<false, -1>": {"FETCH_SIZE_KB_per_launch": 100.0, "WRITE_SIZE_KB_per_launch": 10.0, "launches": 6,
                                                 "SQ_VALU_MFMA_BUSY_CYCLES": 512.0, "GRBM_GUI_ACTIVE": 8.0, "mfma_busy_frac": 0.5},
        "void gemm_rows_x6w_kernel<false>": {"FETCH_SIZE_KB_per_launch": 500.0, "WRITE_SIZE_KB_per_launch": 50.0, "launches": 2,
                                              "SQ_VALU_MFMA_BUSY_CYCLES": 2048.0, "GRBM_GUI_ACTIVE": 16.0, "mfma_busy_frac": 1.0},
        "unrelated": {"FETCH_SIZE_KB_per_launch": 1.0, "launches": 1}}}))
    rec = bench._pmc_record(str(f), "msg_gemm_x6")
    assert rec["launches"] == 8
    assert abs(rec["FETCH_SIZE_KB_per_launch"] - (6 * 100.0 + 2 * 500.0) / 8) < 1e-9
    assert abs(rec["WRITE_SIZE_KB_per_launch"] - (6 * 10.0 + 2 * 50.0) / 8) < 1e-9
    busy, gui = (6 * 512.0 + 2 * 2048.0) / 8, (6 * 8.0 + 2 * 16.0) / 8
    assert abs(rec["mfma_busy_frac"] - busy / (gui / 8.0 * 1024.0)) < 1e-12
    assert bench._pmc_record(str(f), "no_such_kind") is None


def test_box_calibration_fields_reach_the_rooflines():
    """bench.calibrate_rooflines: every roofline of the line gets `frac_calibrated` against THIS box's ceiling -- the power-limited
    bf16 MFMA loop / 3 for an f16x3 kernel, / 6 for a bf16x6 one, the copy rate for an HBM-bound kind -- and `box` the headline per
    calibrated unit and at the reference step clock."""
    import bench

    box = {"mfma_calib_tflops": 1800.0, "hbm_calib_tbs": 4.5, "sclk_mhz_step": 2000.0}
    h3 = {"bound": "mfma", "achieved": 300.0, "peak_basis": "dense fp16 MFMA peak 2500 TF/s / 3 (f16x3: three fp16 MFMA terms per fp32-accurate product)"}
    x6 = {"bound": "mfma", "achieved": 150.0, "peak_basis": "dense bf16 MFMA peak 2500 TF/s / 6 (bf16x6: six bf16 MFMA terms per fp32-accurate product)"}
    hbm = {"bound": "hbm", "achieved": 2250.0, "peak_basis": "HBM3E peak"}
    f32 = {"bound": "mfma", "achieved": 100.0, "peak_basis": "dense fp32 MFMA peak"}
    bench.calibrate_rooflines(box, h3, {"a": {"roofline": x6}, "b": {"roofline": hbm}, "c": {"roofline": f32}, "d": {"roofline": None}}, 4000.0)
    assert h3["peak_calibrated"] == 600.0 and h3["frac_calibrated"] == 0.5
    assert x6["peak_calibrated"] == 300.0 and x6["frac_calibrated"] == 0.5
    assert hbm["peak_calibrated"] == 4500.0 and hbm["frac_calibrated"] == 0.5
    assert "frac_calibrated" not in f32
    assert abs(box["value_per_calibrated_pflops"] - 4000.0 / 1.8) < 0.1
    assert box["ref_sclk_mhz"] == bench.REF_SCLK_MHZ and abs(box["value_at_ref_sclk"] - 4000.0 * (2250.0 / 2000.0) ** 0.4) < 0.1
    assert abs(box["mfma_calib_frac_of_paper_peak"] - 0.72) < 1e-9


def test_smi_sampler_degrades_to_none_without_a_gpu():
    """hip_ops.calibration.SmiSampler: no librocm_smi64 device (the build container) -> summary() is None, nothing raises."""
    import time

    sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
    from buglab.models.hip_ops.calibration import SmiSampler

    with SmiSampler() as s:
        time.sleep(0.01)
    assert s.summary() is None or set(s.summary()) >= {"sclk_mhz", "power_w", "samples"}


def test_roofline_reports_the_ceiling_that_binds():
    """A GEMM kind with known algorithmic bytes is priced against both ceilings; `bound` names the one whose minimum time is larger.
    With f16x3 (833 TF/s) the routed input gradient of configs[1] (80 FLOP/B) is below the 104 FLOP/B ridge: HBM-bound."""
    import bench

    a = argparse.Namespace(model="gnn-mlp", hidden=128, layers=8, nodes=2000, messages=10000, graphs=64, types=16)
    flop = sum(2.0 * 640000 * (2 * d) * d for d in [128] * 6 + [256] * 2)
    kern = {"msg_dgrad_h3": {"launches": 8, "ms": 2.8, "flop": flop, "overlapped": False},
            "msg_wgrad_h3": {"launches": 8, "ms": 2.0, "flop": flop, "overlapped": False},
            "segment_max_ln": {"launches": 8, "ms": 1.6, "flop": 0.0, "overlapped": False}}
    kern = bench.attach_message_gemm_bytes(kern, a, 1)
    assert abs(kern["msg_dgrad_h3"]["alg_bytes"] - 7.36e9) < 0.05e9 and abs(kern["msg_wgrad_h3"]["alg_bytes"] - 1.5e9) < 0.05e9
    r = bench.build_roofline(kern, {}, 1, 0.014, 4800.0, 9.83e9, 19.5e6, brief=True)
    assert r["kernel"] == "msg_dgrad_h3" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - 7.36e9 / 2.8e-3 / 1e9) < 20 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    both = r["both_ceilings"]
    assert both["flop_per_byte"] < both["ridge_flop_per_byte"] and both["frac_of_hbm_ceiling"] > both["frac_of_mfma_ceiling"]
    # the weight gradient (390 FLOP/B) stays MFMA-bound
    del kern["msg_dgrad_h3"]
    r2 = bench.build_roofline(kern, {}, 1, 0.014, 4800.0, 9.83e9, 19.5e6, brief=True)
    assert r2["kernel"] == "msg_wgrad_h3" and r2["bound"] == "mfma" and r2["peak"] == round(bench.MFMA_H3_PEAK_TFLOPS, 1)


def test_dataflow_bytes_of_a_step_add_up():
    """bench.py's whole-step HBM view (`step_frac_of_hbm_roofline_dataflow_bytes`): the bytes a training step's dataflow moves are the sum of
    its kinds, the three message-GEMM kinds are message_gemm_bytes_per_step's, the two E-sized fp32 intermediates are counted written AND
    read, and the figure at BASELINE configs[1] is the 35.9 GB DESIGN.md section 5 quotes."""
    import bench

    H, layers, N, E, T = 128, 8, 64 * 2000, 64 * 10000, 16
    total, kinds = bench.dataflow_bytes_per_step(H, layers, N, E, T, 4.0)
    assert abs(total - sum(kinds.values())) < 1.0
    msg = bench.message_gemm_bytes_per_step(H, layers, N, E, T, 4.0)
    assert kinds["msg_gemm"] == msg["fwd"] and kinds["msg_dgrad"] == msg["dgrad"] and kinds["msg_wgrad"] == msg["wgrad"]
    assert 35.5e9 < total < 36.2e9
    # [E, Dm] messages: written by the forward GEMM, read by the segmented max; [E, 2 Din] input-gradient rows: written by the routed
    # GEMM, read by the segmented sums (six layers (H, H), two (2H, 2H))
    messages = sum(4.0 * E * (2 * H if li % 4 == 3 else H) for li in range(layers))
    rows = sum(4.0 * E * 2 * (2 * H if li % 4 == 3 else H) for li in range(layers))
    assert kinds["msg_gemm"] > messages and kinds["segment_max_ln"] > messages
    assert kinds["msg_dgrad"] > rows and kinds["node_grad_sums"] > rows
    # the bf16x6 split packs 6 instead of 4 bytes per operand element: more bytes, same structure
    total6, _ = bench.dataflow_bytes_per_step(H, layers, N, E, T, 6.0)
    assert total < total6 < 1.1 * total
