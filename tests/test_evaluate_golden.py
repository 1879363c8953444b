"""`buglab.models.evaluate` against the report the REFERENCE's evaluation loop prints for the same predictions
(tests/golden/evaluate_reports.json.gz, made by tests/golden/make_golden_evaluate.py from
/root/reference/buglab/models/evaluate.py:33-255): every count, per-scout line and all six 100-point curves."""
import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))

from buglab.models.evaluate import EvaluationReport, evaluate_predictions, judge_sample  # noqa: E402

with gzip.open(os.path.join(ROOT, "tests", "golden", "evaluate_reports.json.gz"), "rt") as f:
    FIXTURE = json.load(f)


def _predictions(case):
    return [(d, {int(k): v for k, v in lp}, rw) for d, lp, rw in case["predictions"]]


def _split(report):
    """-> (text lines before the curves, {curve name: array})"""
    head, arrays, lines = [], {}, report.splitlines()
    i = 0
    while i < len(lines):
        line = lines[i]
        if " = np.array(" in line:
            name, _, body = line.partition(" = np.")
            while body.count("(") != body.count(")"):
                i += 1
                body += lines[i]
            arrays[name] = eval(body, {"array": np.array, "nan": np.nan})
        elif not line.startswith("###"):
            head.append(line)
        i += 1
    return head, arrays


@pytest.mark.parametrize("case", FIXTURE["cases"], ids=lambda c: c["name"])
def test_report_equals_the_reference_loop(case):
    flags = case["flags"]
    rep = evaluate_predictions(_predictions(case), flags.get("--assume-buggy", False), flags.get("--eval-only-no-bug", False))
    text = rep.format()
    want_head, want_curves = _split(case["report"])
    got_head, got_curves = _split(text)
    assert got_head == want_head
    assert list(got_curves) == list(want_curves)
    for name, want in want_curves.items():
        np.testing.assert_allclose(got_curves[name], want, rtol=0, atol=1e-8, equal_nan=True, err_msg=name)
    if FIXTURE["numpy"] == np.__version__:  # same array printer: the whole text is identical
        assert text == case["report"]


def test_summary_and_scouts_are_consistent():
    case = FIXTURE["cases"][0]
    rep = evaluate_predictions(_predictions(case))
    s, scouts = rep.summary(), rep.per_scout()
    assert s["num_samples"] == len(case["predictions"])
    assert sum(t for _, t in scouts["localization"].values()) == s["num_samples"]
    assert sum(t for _, t in scouts["repair"].values()) == s["num_buggy_samples"]
    assert sum(ok for ok, _ in scouts["localization"].values()) == s["num_location_correct"]
    assert "NoBug" in scouts["localization"] and "NoBug" not in scouts["repair"]
    c = rep.curves()
    assert all(v.shape == (100,) for v in c.values())


def test_edge_cases():
    empty = EvaluationReport([])
    assert empty.summary()["num_samples"] == 0 and np.isnan(empty.summary()["accuracy"])
    d = {"graph": {"reference_nodes": [3, 3, 9]}, "candidate_rewrite_metadata": [["A", 0], ["A", 0], ["B", 0]], "target_fix_action_idx": 2}
    o = judge_sample(d, {3: -0.1, 9: -3.0, -1: -4.0}, [0.0, 1.0, -5.0])
    assert o.warned and not o.location_correct and o.repair_given_location and not o.repaired and o.scout == "B"
    o = judge_sample(d, {3: -3.0, 9: -0.1, -1: -4.0}, [0.0, 1.0, -5.0])
    assert o.location_correct and o.repaired
    o = judge_sample(d, {3: -3.0, 9: -2.0, -1: -0.2}, [0.0, 1.0, -5.0], assume_buggy=True)
    assert o.warned and o.location_correct and abs(np.exp(o.confidence) - 1 / (1 + np.exp(-1.0))) < 1e-6
    with pytest.raises(AssertionError):
        judge_sample(dict(d, target_fix_action_idx=None), {3: -1.0, -1: -1.0}, [0.0, 0.0, 0.0], assume_buggy=True)
