#!/usr/bin/env python
"""
Usage:
    evaluate.py [options] MODEL_FILENAME TEST_DATA_PATH

Options:
    --assume-buggy             Never predict NO_BUG
    --eval-only-no-bug         Evaluate only NO_BUG samples.
    --limit-num-elements=<num> Limit the number of elements to evaluate on.
    --sequential               Do not parallelize data loading. Makes debugging easier.
    -h --help                  Show this screen.

Counterpart of reference buglab/models/evaluate.py (same positional arguments and the options that
do not need Azure).  Metrics follow reference evaluate.py:60-255: localization accuracy, repair
accuracy (overall and given correct location), bug-detection accuracy and false-warning rate.
"""
import argparse
import math
import sys
from pathlib import Path

if __package__ in (None, ""):
    sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

import numpy as np
import torch

from buglab.models.gnn import GnnBugLabModel
from buglab.runtime.richpath import RichPath
from buglab.utils.msgpackutils import load_all_msgpack_l_gz


def evaluate_predictions(predictions, assume_buggy: bool = False, eval_only_no_bug: bool = False):
    """reference evaluate.py:60-200 (core metrics)."""
    num_samples = num_location_correct = 0
    num_buggy = num_repaired_correct = num_repaired_given_location_correct = 0
    num_buggy_and_raised_warning = num_non_buggy_and_no_warning = 0
    for datapoint, location_logprobs, rewrite_probs in predictions:
        if assume_buggy:
            location_logprobs = dict(location_logprobs)
            del location_logprobs[-1]
            norm = float(torch.logsumexp(torch.tensor(list(location_logprobs.values())), dim=-1))
            location_logprobs = {p: v - norm for p, v in location_logprobs.items()}
        target_idx = datapoint["target_fix_action_idx"]
        has_bug = target_idx is not None
        if has_bug and eval_only_no_bug:
            continue
        num_samples += 1
        predicted_node_idx = max(location_logprobs, key=lambda k: location_logprobs[k])
        predicted_rewrite_idx, predicted_rewrite_logprob = None, -math.inf
        for rewrite_idx, (rewrite_node_idx, rewrite_logprob) in enumerate(zip(datapoint["graph"]["reference_nodes"], rewrite_probs)):
            if rewrite_node_idx == predicted_node_idx and rewrite_logprob > predicted_rewrite_logprob:
                predicted_rewrite_idx, predicted_rewrite_logprob = rewrite_idx, rewrite_logprob
        if not has_bug:
            if predicted_node_idx == -1:
                num_location_correct += 1
                num_non_buggy_and_no_warning += 1
            continue
        num_buggy += 1
        target_node = datapoint["graph"]["reference_nodes"][target_idx]
        if predicted_node_idx != -1:
            num_buggy_and_raised_warning += 1
        location_correct = predicted_node_idx == target_node
        num_location_correct += int(location_correct)
        if location_correct and predicted_rewrite_idx == target_idx:
            num_repaired_correct += 1
        # repair accuracy given the correct location
        best, best_lp = None, -math.inf
        for rewrite_idx, (rewrite_node_idx, lp) in enumerate(zip(datapoint["graph"]["reference_nodes"], rewrite_probs)):
            if rewrite_node_idx == target_node and lp > best_lp:
                best, best_lp = rewrite_idx, lp
        num_repaired_given_location_correct += int(best == target_idx)
    nb = max(num_buggy, 1)
    return {
        "num_samples": num_samples,
        "localization_accuracy": num_location_correct / max(num_samples, 1),
        "repair_accuracy": num_repaired_correct / nb,
        "repair_accuracy_given_location": num_repaired_given_location_correct / nb,
        "bug_detection_recall": num_buggy_and_raised_warning / nb,
        "no_bug_precision": num_non_buggy_and_no_warning / max(num_samples - num_buggy, 1),
    }


def run(arguments):
    data_path = RichPath.create(arguments["TEST_DATA_PATH"])
    lim = None if arguments["--limit-num-elements"] is None else int(arguments["--limit-num-elements"])
    data = load_all_msgpack_l_gz(data_path, shuffle=True, limit_num_yielded_elements=lim)
    if not torch.cuda.is_available():
        raise RuntimeError("evaluate.py: no ROCm GPU visible; the BugLab hot path has no CPU fallback")
    device = torch.device("cuda")
    model, nn = GnnBugLabModel.restore_model(Path(arguments["MODEL_FILENAME"]), device)
    predictions = model.predict(data, nn, device, parallelize=not arguments["--sequential"])
    metrics = evaluate_predictions(predictions, arguments["--assume-buggy"], arguments["--eval-only-no-bug"])
    for k, v in metrics.items():
        print(f"{k}: {v}")
    return metrics


if __name__ == "__main__":
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("MODEL_FILENAME")
    p.add_argument("TEST_DATA_PATH")
    p.add_argument("--assume-buggy", action="store_true")
    p.add_argument("--eval-only-no-bug", action="store_true")
    p.add_argument("--limit-num-elements", default=None)
    p.add_argument("--sequential", action="store_true")
    ns = p.parse_args()
    run({"MODEL_FILENAME": ns.MODEL_FILENAME, "TEST_DATA_PATH": ns.TEST_DATA_PATH, "--assume-buggy": ns.assume_buggy,
         "--eval-only-no-bug": ns.eval_only_no_bug, "--limit-num-elements": ns.limit_num_elements, "--sequential": ns.sequential})
