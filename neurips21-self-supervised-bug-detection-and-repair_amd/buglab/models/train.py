#!/usr/bin/env python
"""
Usage:
    train.py [options] MODEL_NAME TRAIN_DATA_PATH VALID_DATA_PATH MODEL_FILENAME

Options:
    --amp                         Use AMP (message GEMMs with fp16 operands, fp32 accumulation)
    --max-num-epochs=<epochs>     The maximum number of epochs to run training for. [default: 100]
    --max-files-per-fold=<n>      The maximum number of files to include in each fold.
    --minibatch-size=<size>       The minibatch size. [default: 300]
    --validate-after=<n_samples>  Run the validation after seen n_samples. [default: 1000000]
    --restore-path=<path>         The path to previous model file for starting from previous checkpoint.
    --sequential                  Do not parallelize data loading. Makes debugging easier.
    --quiet                       Do not show progress bar.
    --model-spec=<json>           Extra model kwargs as JSON (e.g. '{"hidden_state_size": 256}').
    -h --help                     Show this screen.
    --debug                       Enable debug routines. [default: False]

Same command line as reference buglab/models/train.py:1-19 (the Azure-only flags --aml/--azure-info are
dropped: no network).  Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N
--master-addr 127.0.0.1 ... train.py ...`; each rank trains on its share of the stream and gradients are
all-reduced over RCCL once per step (buglab.runtime.trainer).
"""
import argparse
import json
import logging
import os
import sys
from pathlib import Path
from typing import Callable, Iterator, Optional

if __package__ in (None, ""):  # executed as a script, like the reference
    sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from buglab.models.modelregistry import load_model
from buglab.models.utils import LinearWarmupScheduler, optimizer
from buglab.runtime import distributed as D
from buglab.runtime.richpath import RichPath, run_and_debug
from buglab.runtime.shardloader import ShardDataset
from buglab.runtime.trainer import LazyDataIterable, ModelTrainer
from buglab.utils.msgpackutils import load_all_msgpack_l_gz

LOGGER = logging.getLogger(__name__)


def construct_data_loading_callable(data_path: RichPath, shuffle: bool = False, max_files_per_fold: Optional[int] = None,
                                    limit_num_yielded_elements: Optional[int] = None) -> Callable[[], Iterator]:
    return lambda: load_all_msgpack_l_gz(data_path, shuffle=shuffle, take_only_first_n_files=max_files_per_fold,
                                         limit_num_yielded_elements=limit_num_yielded_elements)


def run(arguments):
    """reference train.py:54-138."""
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s %(message)s")
    if os.environ.get("WORLD_SIZE", "1") != "1":
        D.init_from_env("cuda")
    max_files_per_fold = arguments["--max-files-per-fold"]
    if max_files_per_fold is not None:
        max_files_per_fold = int(max_files_per_fold)
    training_data_path = RichPath.create(arguments["TRAIN_DATA_PATH"])
    # same datapoints as the reference's LazyDataIterable(load_all_msgpack_l_gz(...)) (train.py:76-91), plus the
    # list of shard files so that reading + tensorising can run in worker processes (runtime/shardloader.py)
    training_data = ShardDataset(training_data_path, shuffle=True, take_only_first_n_files=max_files_per_fold,
                                 limit_num_yielded_elements=int(arguments["--validate-after"]))
    validation_data = ShardDataset(RichPath.create(arguments["VALID_DATA_PATH"]), take_only_first_n_files=max_files_per_fold)
    model_path = Path(arguments["MODEL_FILENAME"])
    model_spec = {"modelName": arguments["MODEL_NAME"]}
    if arguments.get("--model-spec"):
        model_spec.update(json.loads(arguments["--model-spec"]))
    model, nn, initialize_metadata = load_model(model_spec, model_path, arguments.get("--restore-path", None))
    trainer = ModelTrainer(model, model_path, max_num_epochs=int(arguments["--max-num-epochs"]),
                           minibatch_size=int(arguments["--minibatch-size"]), optimizer_creator=optimizer,
                           clip_gradient_norm=0.5, scheduler_creator=lambda o: LinearWarmupScheduler(o),
                           enable_amp=arguments["--amp"])
    if nn is not None:
        trainer.neural_module = nn
        # continuing from a checkpoint: Adam's moments / warm-up position live in `<checkpoint>.optim` when this
        # trainer wrote it (the model pickle itself stays what the reference's callers expect)
        restored_from = arguments.get("--restore-path") or model_path
        trainer.restore_optimizer_state_from = Path(restored_from)
    trainer.register_train_epoch_end_hook(lambda model, nn, epoch, metrics: LOGGER.info("train epoch %s: %s", epoch, metrics))
    trainer.register_validation_epoch_end_hook(lambda model, nn, epoch, metrics: LOGGER.info("valid epoch %s: %s", epoch, metrics))
    if initialize_metadata:
        import torch
        import torch.distributed as dist

        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            data_for_metadata = LazyDataIterable(construct_data_loading_callable(training_data_path, shuffle=True,
                                                                                  limit_num_yielded_elements=250_000))
            trainer.load_metadata_and_create_network(data_for_metadata, not arguments["--sequential"], not arguments["--quiet"])
        if multi:
            # ONE metadata pass (rank 0) and one set of initial weights for every replica: vocabularies built from
            # differently shuffled samples would map tokens to different embedding rows on different ranks
            box = [(trainer.model, trainer.neural_module) if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            if dist.get_rank() != 0:
                trainer.model, trainer.neural_module = box[0]
    trainer.train(training_data, validation_data, show_progress_bar=not arguments["--quiet"], initialize_metadata=False,
                  parallelize=not arguments["--sequential"], use_multiprocessing=not arguments["--sequential"], patience=10)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("MODEL_NAME")
    p.add_argument("TRAIN_DATA_PATH")
    p.add_argument("VALID_DATA_PATH")
    p.add_argument("MODEL_FILENAME")
    p.add_argument("--amp", action="store_true")
    p.add_argument("--max-num-epochs", default="100")
    p.add_argument("--max-files-per-fold", default=None)
    p.add_argument("--minibatch-size", default="300")
    p.add_argument("--validate-after", default="1000000")
    p.add_argument("--restore-path", default=None)
    p.add_argument("--sequential", action="store_true")
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--model-spec", default=None)
    p.add_argument("--debug", action="store_true")
    ns = p.parse_args(argv)
    d = {"MODEL_NAME": ns.MODEL_NAME, "TRAIN_DATA_PATH": ns.TRAIN_DATA_PATH, "VALID_DATA_PATH": ns.VALID_DATA_PATH,
         "MODEL_FILENAME": ns.MODEL_FILENAME}
    for k, v in vars(ns).items():
        if k.isupper():
            continue
        d["--" + k.replace("_", "-")] = v
    return d


if __name__ == "__main__":
    args = parse_args()
    run_and_debug(lambda: run(args), args.get("--debug", False))
