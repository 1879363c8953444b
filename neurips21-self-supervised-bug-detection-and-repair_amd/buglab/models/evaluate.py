#!/usr/bin/env python
"""
Usage:
    evaluate.py [options] MODEL_FILENAME TEST_DATA_PATH

Options:
    --assume-buggy             Never predict NO_BUG
    --eval-only-no-bug         Evaluate only NO_BUG samples.
    --limit-num-elements=<num> Limit the number of elements to evaluate on.
    --sequential               Do not parallelize data loading. Makes debugging easier.
    --minibatch-size=<size>    Accepted for command-line compatibility (the reference parses and ignores it too).
    --restore-path=<path>      Accepted for command-line compatibility (unused by the reference's run()).
    --quiet                    Accepted for command-line compatibility.
    --debug                    Accepted for command-line compatibility.
    -h --help                  Show this screen.

Counterpart of reference buglab/models/evaluate.py (same positional arguments and the options that
do not need Azure).  The metrics and the printed report follow reference evaluate.py:60-255 line for line in content:
joint / detection / localization / repair-given-location accuracies, the per-scout breakdowns, and the six
threshold curves (false-discovery rate, detection precision / recall, NO_BUG precision, detect-and-repair precision /
recall) sampled at 100 thresholds.  tests/test_evaluate_golden.py compares the report with the one the reference's
own loop prints for the same predictions.
"""
import argparse
import math
import sys
from collections import defaultdict
from pathlib import Path
from typing import Dict, Iterable, List, NamedTuple, Optional

if __package__ in (None, ""):
    sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

import numpy as np
import torch

from buglab.models.gnn import GnnBugLabModel
from buglab.runtime.richpath import RichPath
from buglab.utils.msgpackutils import load_all_msgpack_l_gz

NO_BUG_NODE = -1  # key of the "no bug" option in a sample's location log-probabilities
CURVE_POINTS = 100


class SampleOutcome(NamedTuple):
    """What the report needs from one evaluated sample.  Field order = the sort order of the threshold curves
    (most confident prediction first; ties fall through to the flags, like the reference's tuple sort)."""
    confidence: float                    # log-probability of the predicted location (or of "no bug")
    has_bug: bool
    warned: bool                         # the prediction is a location, not "no bug"
    location_correct: bool               # for a correct sample: "no bug" was predicted
    repair_given_location: Optional[bool]  # best rewrite AT THE TRUE location is the target; None without a bug
    repaired: bool                       # location and rewrite both right
    scout: str                           # rewrite scout of the target, "NoBug" for a correct sample


def _best_rewrite_at(node: int, reference_nodes, rewrite_logprobs) -> Optional[int]:
    best, best_lp = None, -math.inf
    for i, (at, lp) in enumerate(zip(reference_nodes, rewrite_logprobs)):
        if at == node and lp > best_lp:
            best, best_lp = i, lp
    return best


def judge_sample(datapoint, location_logprobs: Dict[int, float], rewrite_logprobs, assume_buggy: bool = False) -> SampleOutcome:
    """One `(datapoint, location_logprobs, rewrite_logprobs)` triple of `model.predict` -> SampleOutcome
    (reference evaluate.py:60-140)."""
    if assume_buggy:  # the "no bug" option is taken away and the rest renormalised (:61-64)
        location_logprobs = {k: v for k, v in location_logprobs.items() if k != NO_BUG_NODE}
        norm = float(torch.logsumexp(torch.tensor(list(location_logprobs.values())), dim=-1))
        location_logprobs = {k: v - norm for k, v in location_logprobs.items()}
    target = datapoint["target_fix_action_idx"]
    has_bug = target is not None
    assert has_bug or not assume_buggy
    reference_nodes = datapoint["graph"]["reference_nodes"]
    predicted_node = max(location_logprobs, key=lambda k: location_logprobs[k])
    predicted_rewrite = _best_rewrite_at(predicted_node, reference_nodes, rewrite_logprobs)
    if has_bug:
        true_node = reference_nodes[target]
        scout = datapoint["candidate_rewrite_metadata"][target][0]
        repair_given_location = _best_rewrite_at(true_node, reference_nodes, rewrite_logprobs) == target
    else:
        true_node, scout, repair_given_location = NO_BUG_NODE, "NoBug", None
    location_correct = predicted_node == true_node
    return SampleOutcome(location_logprobs[predicted_node], has_bug, predicted_node != NO_BUG_NODE, location_correct,
                         repair_given_location, location_correct and predicted_rewrite == target, scout)


class EvaluationReport:
    """Aggregates SampleOutcomes; `summary()` = the scalar metrics, `curves()` = the threshold curves, `format()` =
    the text the reference prints."""

    def __init__(self, outcomes: Iterable[SampleOutcome], eval_only_no_bug: bool = False):
        self.outcomes: List[SampleOutcome] = [o for o in outcomes if not (eval_only_no_bug and o.has_bug)]

    # ---- counts ---------------------------------------------------------------------------------
    def _count(self, pred) -> int:
        return sum(1 for o in self.outcomes if pred(o))

    def summary(self) -> Dict[str, float]:
        n = len(self.outcomes)
        buggy = self._count(lambda o: o.has_bug)
        correct_code = n - buggy
        warned_buggy = self._count(lambda o: o.has_bug and o.warned)
        silent_correct = self._count(lambda o: not o.has_bug and not o.warned)
        div = lambda a, b: a / b if b else float("nan")
        return {
            "num_samples": n,
            "num_buggy_samples": buggy,
            "num_repaired_correct": self._count(lambda o: o.repaired),
            "num_location_correct": self._count(lambda o: o.location_correct),
            "num_repaired_given_location_correct": self._count(lambda o: o.repair_given_location is True),
            "num_detection_correct": warned_buggy + silent_correct,
            "accuracy": div(self._count(lambda o: o.repaired), n),
            "bug_detection_accuracy": div(warned_buggy + silent_correct, n),
            "bug_detection_false_negatives": 1 - div(warned_buggy, buggy),
            "bug_detection_false_positives": 1 - div(silent_correct, correct_code),
            "localization_accuracy": div(self._count(lambda o: o.location_correct), n),
            "repair_accuracy_given_location": div(self._count(lambda o: o.repair_given_location is True), buggy),
            # kept from the first version of this module
            "repair_accuracy": div(self._count(lambda o: o.repaired and o.has_bug), buggy),
            "bug_detection_recall": div(warned_buggy, buggy),
            "no_bug_precision": div(silent_correct, correct_code),
        }

    def per_scout(self) -> Dict[str, Dict[str, List[int]]]:
        """{"localization" | "repair": {scout: [correct, total]}}; repair counts buggy samples only (:104-108,114-118)."""
        loc, rep = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
        for o in self.outcomes:
            loc[o.scout][0] += int(o.location_correct)
            loc[o.scout][1] += 1
            if o.has_bug:
                rep[o.scout][0] += int(bool(o.repair_given_location))
                rep[o.scout][1] += 1
        return {"localization": dict(sorted(loc.items())), "repair": dict(sorted(rep.items()))}

    # ---- threshold curves -------------------------------------------------------------------------
    def curves(self) -> Dict[str, np.ndarray]:
        """Samples ordered by decreasing confidence; every curve is a running ratio over that order, resampled at
        CURVE_POINTS probability thresholds in [0, 1] (:186-255).  Kept as the reference has it: a warning counts as
        false for the detect-and-repair curves when the rewrite at the TRUE location is wrong, and a warning on
        correct code counts as a wrong location; NO_BUG precision is 1 - false alarms / (number of buggy samples)."""
        ranked = sorted(self.outcomes, key=lambda o: tuple(o[:5]), reverse=True)
        col = lambda f: np.array([f(o) for o in ranked], dtype=bool)
        has_bug, warned, loc_ok = col(lambda o: o.has_bug), col(lambda o: o.warned), col(lambda o: o.location_correct)
        repair_ok = col(lambda o: bool(o.repair_given_location))
        num_buggy = int(has_bug.sum())
        det_true = np.cumsum(has_bug & loc_ok)
        det_false = np.cumsum(warned & ~loc_ok)
        full_true = np.cumsum(has_bug & loc_ok & repair_ok)
        full_false = np.cumsum(warned & (~loc_ok | ~repair_ok))
        false_alarms = np.cumsum(warned & ~has_bug)
        with np.errstate(divide="ignore", invalid="ignore"):
            running = {
                "fdr": det_false / (det_true + det_false),
                "detection_precision": det_true / (det_true + det_false),
                "detection_recall": det_true / num_buggy,
                "no_bug_precision": 1 - false_alarms / (num_buggy + 1e-10),
                "precision": full_true / (full_true + full_false),
                "recall": full_true / num_buggy,
            }
        x = np.linspace(0, 1, num=CURVE_POINTS)
        prob = np.exp(np.array([o.confidence for o in ranked]))
        out = {"x": x}
        for name, values in running.items():  # np.interp wants increasing abscissae: walk the ranking backwards
            out[name] = np.interp(x, prob[::-1], values[::-1], right=0)
        return out

    # ---- text -------------------------------------------------------------------------------------
    def format(self) -> str:
        s, scouts = self.summary(), self.per_scout()
        n, buggy = s["num_samples"], s["num_buggy_samples"]
        bar, lines = "=" * 34, []
        add = lines.append
        add(bar)
        add(f"Accuracy (Localization & Repair) {s['accuracy']:.2%} ({s['num_repaired_correct']}/{n})")
        add(f"Bug Detection Accuracy (no Localization or Repair) {s['bug_detection_accuracy']:.2%} ({s['num_detection_correct']}/{n})")
        fn = f"{s['bug_detection_false_negatives']:.2%}" if buggy > 0 else "NaN (0/0)"
        fp = f"{s['bug_detection_false_positives']:.2%}" if n - buggy > 0 else "NaN (0/0)"
        add(f"Bug Detection (no Localization or Repair) False Negatives: {fn}")
        add(f"Bug Detection (no Localization or Repair) False Positives: {fp}")
        add(bar)
        add(f"Localization Accuracy {s['localization_accuracy']:.2%} ({s['num_location_correct']}/{n})")
        for scout, (ok, total) in scouts["localization"].items():
            add(f"\t{scout}: {ok / total:.1%}  ({ok}/{total})")
        add("=" * 41)
        if buggy == 0:
            add("--eval-only-no-bug is True. Repair Accuracy Given Location cannot be computed.")
        else:
            add(f"Repair Accuracy Given Location {s['repair_accuracy_given_location']:.2%}  ({s['num_repaired_given_location_correct']}/{buggy})")
        for scout, (ok, total) in scouts["repair"].items():
            add(f"\t{scout}: {ok / total:.1%}  ({ok}/{total})")
        c = self.curves()
        add("x = np." + repr(c["x"]))
        for title, name in (("False Detection Rate", "fdr"), ("Detection Precision", "detection_precision"),
                            ("Detection Recall", "detection_recall"), ("Detection NO_BUG Precision", "no_bug_precision"),
                            ("Precision (Detect and Repair)", "precision"), ("Recall (Detect and Repair)", "recall")):
            add(f"### {title} ###")
            add(f"{name} = np." + repr(c[name]))
        return "\n".join(lines) + "\n"


def evaluate_predictions(predictions, assume_buggy: bool = False, eval_only_no_bug: bool = False) -> EvaluationReport:
    """`predictions` = what `model.predict` yields.  With `eval_only_no_bug` the buggy samples are skipped before
    anything is computed for them (:69-70)."""
    outcomes = []
    for datapoint, location_logprobs, rewrite_logprobs in predictions:
        if eval_only_no_bug and datapoint["target_fix_action_idx"] is not None:
            continue
        outcomes.append(judge_sample(datapoint, location_logprobs, rewrite_logprobs, assume_buggy))
    return EvaluationReport(outcomes)


def run(arguments):
    data_path = RichPath.create(arguments["TEST_DATA_PATH"])
    lim = None if arguments["--limit-num-elements"] is None else int(arguments["--limit-num-elements"])
    data = load_all_msgpack_l_gz(data_path, shuffle=True, limit_num_yielded_elements=lim)
    if not torch.cuda.is_available():
        raise RuntimeError("evaluate.py: no ROCm GPU visible; the BugLab hot path has no CPU fallback")
    device = torch.device("cuda")
    model, nn = GnnBugLabModel.restore_model(Path(arguments["MODEL_FILENAME"]), device)
    predictions = model.predict(data, nn, device, parallelize=not arguments["--sequential"])
    report = evaluate_predictions(predictions, arguments["--assume-buggy"], arguments["--eval-only-no-bug"])
    sys.stdout.write(report.format())
    return report.summary()


if __name__ == "__main__":
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("MODEL_FILENAME")
    p.add_argument("TEST_DATA_PATH")
    p.add_argument("--assume-buggy", action="store_true")
    p.add_argument("--eval-only-no-bug", action="store_true")
    p.add_argument("--limit-num-elements", default=None)
    p.add_argument("--sequential", action="store_true")
    p.add_argument("--minibatch-size", default=300)
    p.add_argument("--restore-path", default=None)
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--debug", action="store_true")
    ns = p.parse_args()
    run({"MODEL_FILENAME": ns.MODEL_FILENAME, "TEST_DATA_PATH": ns.TEST_DATA_PATH, "--assume-buggy": ns.assume_buggy,
         "--eval-only-no-bug": ns.eval_only_no_bug, "--limit-num-elements": ns.limit_num_elements, "--sequential": ns.sequential})
