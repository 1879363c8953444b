#!/usr/bin/env python
"""
Usage:
    trainandeval.py [options] MODEL_NAME TRAIN_DATA_PATH VALID_DATA_PATH TEST_DATA_PATH MODEL_FILENAME

Options:
    --amp                         Use AMP (message GEMMs with fp16 operands, fp32 accumulation)
    --limit-num-elements=<num>    Limit the number of elements to evaluate on.
    --max-num-epochs=<epochs>     The maximum number of epochs to run training for. [default: 100]
    --max-files-per-fold=<n>      The maximum number of files to include in each fold.
    --minibatch-size=<size>       The minibatch size. [default: 300]
    --restore-path=<path>         The path to previous model file for starting from previous checkpoint.
    --validate-after=<n_samples>  Run the validation after seen n_samples. [default: 1000000]
    --sequential                  Do not parallelize data loading. Makes debugging easier.
    --quiet                       Do not show progress bar.
    --model-spec=<json>           Extra model kwargs as JSON.
    -h --help                     Show this screen.
    --debug                       Enable debug routines. [default: False]

Counterpart of reference buglab/models/trainandeval.py:1-29: train.run(args), then evaluate.run(args) on TEST_DATA_PATH
with the model that training wrote to MODEL_FILENAME -- one argument dictionary for both, as in the reference (the
Azure-only flags --aml / --azure-info are dropped: no network).
"""
import argparse
import sys
from pathlib import Path

if __package__ in (None, ""):  # executed as a script, like the reference
    sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from buglab.models import evaluate, train
from buglab.runtime.richpath import run_and_debug


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name in ("MODEL_NAME", "TRAIN_DATA_PATH", "VALID_DATA_PATH", "TEST_DATA_PATH", "MODEL_FILENAME"):
        p.add_argument(name)
    p.add_argument("--amp", action="store_true")
    p.add_argument("--limit-num-elements", default=None)
    p.add_argument("--max-num-epochs", default="100")
    p.add_argument("--max-files-per-fold", default=None)
    p.add_argument("--minibatch-size", default="300")
    p.add_argument("--restore-path", default=None)
    p.add_argument("--validate-after", default="1000000")
    p.add_argument("--sequential", action="store_true")
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--model-spec", default=None)
    p.add_argument("--debug", action="store_true")
    ns = p.parse_args(argv)
    d = {k: v for k, v in vars(ns).items() if k.isupper()}
    for k, v in vars(ns).items():
        if not k.isupper():
            d["--" + k.replace("_", "-")] = v
    # options evaluate.run() reads that this command line does not offer (reference evaluate.py:1-18 defaults)
    d.setdefault("--assume-buggy", False)
    d.setdefault("--eval-only-no-bug", False)
    return d


def main(argv=None):
    args = parse_args(argv)
    run_and_debug(lambda: train.run(args), args.get("--debug", False))
    return run_and_debug(lambda: evaluate.run(args), args.get("--debug", False))


if __name__ == "__main__":
    main()
