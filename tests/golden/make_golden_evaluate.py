#!/usr/bin/env python
"""Golden report of the REFERENCE's evaluation loop (buglab/models/evaluate.py:33-255), run unmodified from
/root/reference (build container only) on synthetic predictions:

    python tests/golden/make_golden_evaluate.py       # rewrites tests/golden/evaluate_reports.json.gz

`run()` in the reference is one function from "load model" to "print curves"; here the model and the data loader are
replaced (a fake `restore_model` whose `predict` replays stored `(datapoint, location_logprobs, rewrite_probs)`
triples) and what it prints is captured.  The fixture holds the triples and the printed report for three argument
sets: default, `--assume-buggy` (on the buggy samples) and `--eval-only-no-bug`.
"""
import contextlib
import gzip
import io
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import make_golden as MG  # noqa: E402

SCOUTS = ["VariableMisuse", "ArgSwap", "BinaryOperator", "Literal"]


def make_predictions(n, seed, only_buggy=False):
    """What `predict` yields: the datapoint (fields the metric loop reads), {node id | -1: log-probability}, and one
    log-probability per candidate rewrite.  Log-probabilities are rounded to 4 decimals and made unique per set so
    that neither ties nor float formatting can blur the comparison."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        n_loc = int(rng.integers(2, 7))
        loc_nodes = sorted(rng.choice(np.arange(5, 80), size=n_loc, replace=False).tolist())
        reference_nodes, metadata = [], []
        for node in loc_nodes:
            for _ in range(int(rng.integers(1, 4))):
                reference_nodes.append(int(node))
                metadata.append([SCOUTS[int(rng.integers(0, len(SCOUTS)))], "rewrite"])
        buggy = only_buggy or rng.random() < 0.6
        target = int(rng.integers(0, len(reference_nodes))) if buggy else None
        logits = rng.normal(size=n_loc + 1) * 2.0
        if buggy and rng.random() < 0.6:  # a model that is often right
            logits[loc_nodes.index(reference_nodes[target])] += 3.0
        if not buggy and rng.random() < 0.6:
            logits[-1] += 3.0
        lp = logits - np.log(np.exp(logits).sum())
        lp = np.round(lp, 4) - 1e-6 * (i + 1)
        location_logprobs = {int(k): float(v) for k, v in zip(loc_nodes + [-1], lp)}
        rw = np.round(rng.normal(size=len(reference_nodes)), 4) + 1e-5 * np.arange(len(reference_nodes))
        if buggy and rng.random() < 0.7:
            rw[target] += 2.5
        datapoint = {"graph": {"reference_nodes": reference_nodes}, "candidate_rewrite_metadata": metadata, "target_fix_action_idx": target}
        out.append([datapoint, location_logprobs, [float(x) for x in rw]])
    return out


def main():
    MG._install_stubs()
    sys.modules["docopt"] = types.SimpleNamespace(docopt=lambda *_a, **_k: {})
    sys.modules["dpu_utils.utils"].run_and_debug = lambda f, _d: f()
    for name in ("msgpack",):
        try:
            __import__(name)
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import buglab.models.evaluate as ref_eval  # noqa: reference code

    cases = []
    for name, preds, flags in (
        ("default", make_predictions(60, 1), {}),
        ("assume_buggy", make_predictions(40, 2, only_buggy=True), {"--assume-buggy": True}),
        ("eval_only_no_bug", make_predictions(50, 3), {"--eval-only-no-bug": True}),
    ):
        fake_model = types.SimpleNamespace(predict=lambda data, nn, device, parallelize, _p=preds: iter([(d, dict(lp), list(rw)) for d, lp, rw in _p]))
        ref_eval.RichPath = types.SimpleNamespace(create=lambda path, azure=None: path)
        ref_eval.load_all_msgpack_l_gz = lambda *a, **k: iter(())
        ref_eval.GnnBugLabModel = types.SimpleNamespace(restore_model=lambda path, device, _m=fake_model: (_m, None))
        args = {"MODEL_FILENAME": "model.pkl.gz", "TEST_DATA_PATH": "test", "--limit-num-elements": None, "--sequential": True}
        args.update(flags)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
            ref_eval.run(args)
        cases.append({"name": name, "flags": flags,
                      "predictions": [[d, [[k, v] for k, v in lp.items()], rw] for d, lp, rw in preds],
                      "report": buf.getvalue()})
        print(f"{name}: {len(preds)} predictions, report of {len(buf.getvalue().splitlines())} lines")
    with gzip.open(os.path.join(OUT, "evaluate_reports.json.gz"), "wt") as f:
        json.dump({"numpy": np.__version__, "cases": cases}, f)


if __name__ == "__main__":
    main()
