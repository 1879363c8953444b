"""`ModelTrainer` -- the training loop the reference drives through ptgnn (contract pinned at
reference buglab/models/train.py:98-134 and buglab/controllers/trainbugdetector.py:73-153): hooks,
`load_metadata_and_create_network`, `train(..., patience=)`, overridable `_run_validation`, the
module invoked as `nn(**minibatch)`.

MI355X specifics: one process per GPU; when `torch.distributed` is initialised each rank consumes
its own share of the tensorised stream, gradients are summed with ONE RCCL all-reduce of the flat
gradient buffer per step (buglab.runtime.optim.FlatAdam) weighted by B_rank / B_total so the update
equals the single-process full-minibatch update; only rank 0 saves checkpoints."""
from __future__ import annotations

import logging
import os
import time
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from buglab.runtime import distributed as D
from buglab.runtime.optim import FlatAdam

LOGGER = logging.getLogger(__name__)


def _record_stream(obj, stream) -> None:
    """Every device tensor reachable from a (nested) minibatch is marked as in use on `stream`."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


def _prefetch(iterator, depth: int = 2):
    """Runs `iterator` in a background thread, `depth` items ahead; exceptions are re-raised in the consumer.
    Closing the returned generator (or dropping it) stops the thread and closes `iterator`, so that a consumer
    that stops early -- a rank whose peers ran out of minibatches -- does not leave loader processes behind."""
    import queue
    import threading

    q: "queue.Queue" = queue.Queue(maxsize=depth)
    end = object()
    stop = threading.Event()

    def put(x) -> bool:
        while not stop.is_set():
            try:
                q.put(x, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for x in iterator:
                if not put(x):
                    break
            else:
                put(end)
        except BaseException as e:
            put(e)
        finally:
            close = getattr(iterator, "close", None)
            if close is not None:
                close()

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            x = q.get()
            if x is end:
                return
            if isinstance(x, BaseException):
                raise x
            yield x
    finally:
        stop.set()
        try:
            while True:
                q.get_nowait()
        except queue.Empty:
            pass
        t.join(timeout=5.0)


class AbstractScheduler:
    def step(self, epoch_idx: int, epoch_step: int) -> None:
        raise NotImplementedError


class LazyDataIterable:
    """iterable-from-callable (reference train.py:76-91)."""

    def __init__(self, base_iterable_func: Callable[[], Iterable]):
        self._f = base_iterable_func

    def __iter__(self):
        return iter(self._f())


class ModelTrainer:
    def __init__(self, model, save_location: Path, *, max_num_epochs: int = 100, minibatch_size: int = 200,
                 optimizer_creator: Optional[Callable] = None, clip_gradient_norm: Optional[float] = None,
                 scheduler_creator: Optional[Callable] = None, target_validation_metric: Optional[str] = None,
                 target_validation_metric_higher_is_better: bool = False, enable_amp: bool = False):
        self.model = model
        self._save_location = Path(save_location)
        self._max_num_epochs = max_num_epochs
        self._minibatch_size = minibatch_size
        self._optimizer_creator = optimizer_creator or (lambda params: FlatAdam(params))
        self._clip = clip_gradient_norm
        self._scheduler_creator = scheduler_creator
        self._target_metric = target_validation_metric
        self._target_higher_better = target_validation_metric_higher_is_better
        # --amp (reference train.py:8,106 -> ptgnn's autocast + GradScaler): on this path the message GEMMs -- half of a step --
        # run with fp16 operands (one MFMA term instead of f16x3's three, half the operand bytes), fp32 accumulation and fp32
        # results; the gradient operand is scaled by its device-side amax, so no GradScaler and no skipped steps.  Everything
        # else stays fp32.  Applied when training starts (`train`), undone when it ends.
        self._enable_amp = bool(enable_amp)
        self._nn = None
        self._use_multiprocessing = False
        self._train_epoch_end_hooks: List[Callable] = []
        self._validation_epoch_end_hooks: List[Callable] = []
        self._training_start_hooks: List[Callable] = []

    # -- contract ---------------------------------------------------------------------------------
    @property
    def neural_module(self):
        if self._nn is None:
            raise Exception("Neural module does not exist. Metadata needs to be loaded first.")
        return self._nn

    @neural_module.setter
    def neural_module(self, nn):
        self._nn = nn

    def register_train_epoch_end_hook(self, hook):
        self._train_epoch_end_hooks.append(hook)

    def register_validation_epoch_end_hook(self, hook):
        self._validation_epoch_end_hooks.append(hook)

    def register_training_start_hook(self, hook):
        self._training_start_hooks.append(hook)

    def load_metadata_and_create_network(self, training_data: Iterable, parallelize: bool = True, show_progress_bar: bool = True):
        self.model.compute_metadata(training_data, parallelize, show_progress_bar)
        self._nn = self.model.build_neural_module()
        LOGGER.info("Model has %s trainable parameters.", sum(p.numel() for p in self._nn.parameters() if p.requires_grad))

    # -- helpers ----------------------------------------------------------------------------------
    @staticmethod
    def _world():
        return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)

    def _rank_share(self, data: Iterable):
        """This rank's datapoints of a stream every rank reads in the SAME order: windows of the stream are split by
        message count (greedy bin-packing, buglab.runtime.distributed.balanced_rank_share) -- a step's time follows the
        number of messages, not the number of graphs."""
        rank, world = self._world()
        yield from D.balanced_rank_share(data, rank, world)

    def _iter_minibatches(self, data, device, parallelize, shuffle_key=None, use_prestarted: bool = False):
        """use_prestarted: only the TRAINING epoch loop passes True -- the pool forked by `_prestart_loaders` belongs to the
        next training epoch; validation runs in between and must neither consume nor close it."""
        from buglab.runtime.shardloader import collated_minibatches_parallel, default_num_workers

        workers = default_num_workers() if (parallelize and self._use_multiprocessing) else 0
        if workers > 0 and hasattr(data, "shard_files") and hasattr(self.model, "collate_minibatch"):
            # shard files are read, tensorised and collated by worker processes; this process copies whole
            # minibatches to the device and drives the GPU
            from buglab.runtime.shardloader import receive_packed

            rank, world = self._world()
            limit = getattr(data, "limit_num_yielded_elements", None)

            self.last_input_timing = timing = {"loader_wait_s": 0.0, "upload_s": 0.0, "minibatches": 0}
            upload_stream = None
            if torch.device(device).type == "cuda":
                # one stream for the trainer's lifetime: a stream object that dies would have its destructor run in
                # the loader processes forked later (they inherit the interpreter's garbage), and HIP does not survive a fork
                if getattr(self, "_upload_stream", None) is None:
                    self._upload_stream = torch.cuda.Stream(device)
                upload_stream = self._upload_stream

            def received():  # runs in a prefetch thread: the staging copy + pinned H2D copy overlap the trainer
                seen = 0     # thread's kernel launches; the int32 blob comes through shared memory, not the pipe
                files = data.shard_files()
                # forked while the previous epoch's validation was running (validation itself never takes it)
                pool = self._take_prestarted_pool(data, files) if use_prestarted else None
                source = collated_minibatches_parallel(self.model, files, workers, self._minibatch_size, rank, world, packed=True, pool=pool)
                try:
                    while True:
                        t0 = time.perf_counter()
                        item = next(source, None)
                        t1 = time.perf_counter()
                        if item is None:
                            break
                        if upload_stream is not None:
                            # the copy runs on its own stream (its own allocator pool): it overlaps the step that is running instead
                            # of queueing behind it; the trainer's stream waits for `ready` before the first kernel of the step
                            with torch.cuda.stream(upload_stream):
                                mb = receive_packed(item, device)
                                ready = torch.cuda.Event()
                                ready.record(upload_stream)
                            mb["_upload_ready"] = ready
                        else:
                            mb = receive_packed(item, device)
                        timing["loader_wait_s"] += t1 - t0
                        timing["upload_s"] += time.perf_counter() - t1
                        timing["minibatches"] += 1
                        yield mb
                        seen += int(item[2]["num_graphs"]) * world
                        if limit is not None and seen >= limit:
                            break
                finally:
                    source.close()

            for mb in _prefetch(received(), depth=3):
                ready = mb.pop("_upload_ready", None)
                if ready is not None:
                    torch.cuda.current_stream().wait_event(ready)
                    # the blob was allocated on the upload stream and is used on this one: its memory may only be recycled
                    # once this stream is done with it
                    _record_stream(mb, torch.cuda.current_stream())
                yield mb
            return
        tensors = self.model.tensorize_dataset(self._rank_share(data), parallelize=parallelize)
        for mb, _ in self.model.minibatch_iterator(tensors, device, self._minibatch_size, parallelize=parallelize):
            yield mb

    # ---- the next training epoch's loader processes are forked ahead of time -------------------------------------------
    def _prestart_loaders(self, data, epoch: int, parallelize: bool) -> None:
        """Fork the loader processes of training epoch `epoch` NOW (the caller is about to run validation): forking 8-32
        processes from a process with the GPU runtime mapped and reading the first shards cost 0.3-0.7 s per epoch when it
        happened at the epoch's first `next()`.  The pool is handed to `_iter_minibatches` when that epoch starts; it is
        closed there, or by `_drop_prestarted_pool` when training stops first."""
        from buglab.runtime.shardloader import default_num_workers, minibatch_pool

        self._drop_prestarted_pool()
        workers = default_num_workers() if (parallelize and self._use_multiprocessing) else 0
        if workers <= 0 or not (hasattr(data, "shard_files") and hasattr(data, "set_epoch") and hasattr(self.model, "collate_minibatch")):
            return
        if os.environ.get("BUGLAB_PRESTART_LOADERS", "1") == "0":
            return
        rank, world = self._world()
        data.set_epoch(epoch)
        files = data.shard_files()
        pool = minibatch_pool(self.model, files, workers, self._minibatch_size, rank, world, packed=True).start()
        self._prestarted = (id(data), tuple(files), pool)

    def _take_prestarted_pool(self, data, files):
        pre = getattr(self, "_prestarted", None)
        if pre is None:
            return None
        self._prestarted = None
        if pre[0] == id(data) and pre[1] == tuple(files):
            return pre[2]
        pre[2].close()  # (different data or file order: not what this epoch reads)
        return None

    def _drop_prestarted_pool(self) -> None:
        pre = getattr(self, "_prestarted", None)
        self._prestarted = None
        if pre is not None:
            pre[2].close()

    @staticmethod
    def _all_ranks_have(flag: bool, device) -> bool:
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return flag
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def _run_training(self, training_data, epoch, device, optimizer, scheduler, parallelize):
        nn = self.neural_module
        nn.train()
        nn.reset_metrics()
        if hasattr(training_data, "set_epoch"):
            training_data.set_epoch(epoch)  # every rank shuffles the shard files with the same per-epoch seed
        it = iter(self._iter_minibatches(training_data, device, parallelize, use_prestarted=True))
        step, num_graphs, t0 = 0, 0, time.time()
        B = 0
        _, world = self._world()
        fused_dp = world > 1 and hasattr(optimizer, "step_data_parallel")
        waited, first_wait = 0.0, None  # time this thread spent blocked on the input pipeline (first minibatch apart)
        try:
            while True:
                tw = time.perf_counter()
                mb = next(it, None)
                tw = time.perf_counter() - tw
                if first_wait is None:
                    first_wait = tw
                else:
                    waited += tw
                if fused_dp:
                    # Ranks stay in lock step through the gradient all-reduce alone: a rank whose loader is exhausted
                    # keeps stepping with an empty contribution until the tail of the all-reduce says that nobody had
                    # a minibatch (read one step late from pinned memory: no device synchronisation, no extra collective)
                    # Only a rank that itself had NOTHING in the previous step can have seen an idle step, and such a rank
                    # has no work queued behind the read -- so the pinned flag is only waited for when waiting is free;
                    # a rank that is still training never blocks on its previous step here.
                    if step > 0 and B == 0 and optimizer.previous_step_was_idle():
                        break
                    optimizer.zero_grad()
                    B = int(mb["has_bug"].shape[0]) if mb is not None else 0
                    if hasattr(optimizer, "begin_data_parallel_step"):
                        optimizer.begin_data_parallel_step(B)  # layer-wise gradient buckets reduce during backward
                    if mb is not None:
                        try:
                            loss = nn(**mb)
                            loss.backward()
                        except BaseException:
                            # the layer-wise all-reduces are armed: disarm them and complete this step's plan before the
                            # exception leaves (a stale callback would issue collectives the peers do not expect)
                            if hasattr(optimizer, "abort_data_parallel_step"):
                                optimizer.abort_data_parallel_step()
                            raise
                    optimizer.step_data_parallel(B)
                else:
                    if not self._all_ranks_have(mb is not None, device):
                        break
                    optimizer.zero_grad()
                    loss = nn(**mb)
                    loss.backward()
                    B = int(mb["has_bug"].shape[0])
                    optimizer.step(D.global_batch_weight(B, device))
                if scheduler is not None:
                    scheduler.step(epoch_idx=epoch, epoch_step=step)
                step += 1
                num_graphs += B
        finally:
            it.close()  # a rank that stops before its loader is exhausted shuts the loader processes down
        metrics = nn.report_metrics()
        elapsed = time.time() - t0
        self._check_f16x3_saturation(device, epoch)
        LOGGER.info("Epoch %s: %s steps, %.1f graphs/s (this rank). Train metrics: %s", epoch, step, num_graphs / max(elapsed, 1e-9), metrics)
        # where the epoch went on the host side: a large share of `input wait` means the loaders, not the device, set the pace
        self.last_epoch_timing = {"elapsed_s": elapsed, "steps": step, "first_minibatch_s": first_wait or 0.0, "input_wait_s": waited}
        LOGGER.info("Epoch %s timing: first minibatch after %.2f s, %.2f s of %.2f s blocked on input", epoch, first_wait or 0.0, waited, elapsed)
        return metrics

    @staticmethod
    def _check_f16x3_saturation(device, epoch) -> None:
        """The f16x3 message GEMMs pack layer inputs and weights into fp16 planes with fixed power-of-two scales (hip_ops,
        csrc/bl_gemm_h3.hip): a value beyond the range saturates instead of overflowing.  Once per epoch the device-side counter
        of such events is read (one synchronising 4-byte copy) -- a healthy run has none; a diverging one should say so."""
        if torch.device(device).type != "cuda":
            return
        from buglab.models import hip_ops

        n = hip_ops.h3_saturation_events(reset=True)
        if n > 0:
            LOGGER.warning("Epoch %s: %s f16x2 packing threads saturated a value (a layer input beyond +-255.9 or a weight beyond +-1023): "
                           "the f16x3 message GEMMs clipped it.  If the run is not diverging, train with the bf16x6 split "
                           "(BL_MSG_GEMM=x6 / hip_ops.set_msg_gemm_mode('bf16x6')).", epoch, n)

    def _run_validation(self, validation_tensors, epoch, best_target_metric, device, parallelize, show_progress_bar):
        """-> (target metric, improved?)   (overridden by the reference's detector trainer,
        trainbugdetector.py:130-143)."""
        nn = self.neural_module
        nn.eval()
        nn.reset_metrics()
        total, n = torch.zeros((), device=device), 0
        with torch.no_grad():
            # validation needs no lock step between ranks: each rank runs through its own share, ONE all-reduce at the end
            for mb in self._iter_minibatches(validation_tensors, device, parallelize):
                total += nn(**mb).detach()
                n += 1
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if distributed:
            t = torch.stack([total, torch.tensor(float(n), device=device)])
            dist.all_reduce(t)
            total, n = t[0], int(t[1].item())
        metrics = nn.report_metrics()  # per-rank counters (the loss above is global)
        val_loss = float(total) / max(n, 1)
        for hook in self._validation_epoch_end_hooks:
            hook(self.model, nn, epoch, metrics)
        if self._target_metric is not None:
            target = metrics[self._target_metric]
            improved = target > best_target_metric if self._target_higher_better else target < best_target_metric
        else:
            target, improved = val_loss, val_loss < best_target_metric
        if distributed:
            # every rank must take the same early-stopping / checkpoint branch: rank 0's decision is the decision
            # (a target metric, or an overridden _run_validation, is computed from rank-local counters)
            decision = [float(target), bool(improved)]
            dist.broadcast_object_list(decision, src=0)
            target, improved = decision
        LOGGER.info("Epoch %s: validation loss %.5f. Metrics: %s", epoch, val_loss, metrics)
        return target, improved

    # ---- optimiser state next to the checkpoint -----------------------------------------------------------------
    # The model pickle is what the reference's callers exchange; Adam's moments and step count (warm-up position) go
    # to a sidecar `<checkpoint>.optim` so that continuing from a checkpoint does not restart them.  Optional on both
    # sides: an optimiser without state_dict() writes nothing, a missing or mismatching sidecar is ignored with a note.
    @staticmethod
    def _optimizer_sidecar(path) -> Path:
        path = Path(path)
        return path.with_name(path.name + ".optim")

    def _save_optimizer_state(self, optimizer) -> None:
        if not hasattr(optimizer, "state_dict"):
            return
        import os

        target = self._optimizer_sidecar(self._save_location)
        tmp = target.with_name(target.name + ".tmp")
        state = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in optimizer.state_dict().items()}
        torch.save(state, tmp)
        os.replace(tmp, target)

    def _restore_optimizer_state(self, optimizer, device) -> None:
        source = getattr(self, "restore_optimizer_state_from", None)
        if source is None or not hasattr(optimizer, "load_state_dict"):
            return
        sidecar = self._optimizer_sidecar(source)
        if not sidecar.exists():
            LOGGER.info("No optimiser state next to %s: moments and warm-up start afresh.", source)
            return
        try:
            optimizer.load_state_dict(torch.load(sidecar, map_location=device, weights_only=False))
            LOGGER.info("Optimiser state restored from %s.", sidecar)
        except Exception as e:  # a checkpoint of another architecture: shapes differ
            LOGGER.warning("Optimiser state in %s does not fit this model (%s): starting afresh.", sidecar, e)

    def train(self, training_data: Iterable, validation_data: Iterable, *, show_progress_bar: bool = True,
              initialize_metadata: bool = True, parallelize: bool = True, use_multiprocessing: bool = False, patience: int = 5,
              device=None):
        self._use_multiprocessing = bool(use_multiprocessing)
        if initialize_metadata:
            self.load_metadata_and_create_network(training_data, parallelize, show_progress_bar)
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("ModelTrainer.train: no ROCm GPU visible; the BugLab hot path has no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self._nn = self.neural_module.to(device)
        optimizer = self._optimizer_creator(self._nn.parameters())
        if self._clip is not None:
            if hasattr(optimizer, "clip"):
                optimizer.clip = self._clip
            else:
                LOGGER.warning("clip_gradient_norm=%s was requested but %s cannot apply it (only FlatAdam fuses the clip); "
                               "gradients are NOT clipped", self._clip, type(optimizer).__name__)
        self._restore_optimizer_state(optimizer, device)
        _, world = self._world()
        if world > 1 and hasattr(optimizer, "set_overlap_groups") and hasattr(self._nn, "overlap_parameter_groups"):
            # the gradient all-reduce goes layer by layer, behind the backward pass (runtime/optim.py::set_overlap_groups)
            optimizer.set_overlap_groups(self._nn.overlap_parameter_groups())
        if world > 1:
            # replicas must be identical before the first step: rank 0's parameters (and moments) win
            if hasattr(optimizer, "broadcast_parameters"):
                optimizer.broadcast_parameters(0)
            else:
                for p in self._nn.parameters():
                    dist.broadcast(p.data, src=0)
                from buglab.models import hip_ops

                hip_ops.invalidate_weight_packs()  # `.data` writes do not bump `_version`: packed / transposed copies are stale
        scheduler = self._scheduler_creator(optimizer) if self._scheduler_creator is not None else None
        for hook in self._training_start_hooks:
            hook(self.model, self._nn, optimizer)
        from buglab.models import hip_ops

        hip_ops.use_step_stream(device)  # the step's chain on a high-priority stream, the weight-gradient GEMMs behind it on a normal one
        rank, _ = self._world()
        best = float("-inf") if (self._target_metric is not None and self._target_higher_better) else float("inf")
        bad_epochs = 0
        prev_gemm_mode = hip_ops.set_msg_gemm_mode("f16x1") if self._enable_amp else None
        if self._enable_amp:
            LOGGER.info("--amp: message GEMMs with fp16 operands (one MFMA term, fp32 accumulation); was %s", prev_gemm_mode)
        try:
            for epoch in range(self._max_num_epochs):
                metrics = self._run_training(training_data, epoch, device, optimizer, scheduler, parallelize)
                for hook in self._train_epoch_end_hooks:
                    hook(self.model, self._nn, epoch, metrics)
                if epoch + 1 < self._max_num_epochs:
                    self._prestart_loaders(training_data, epoch + 1, parallelize)  # they fill their queue during validation
                target, improved = self._run_validation(validation_data, epoch, best, device, parallelize, show_progress_bar)
                if improved:
                    best, bad_epochs = target, 0
                    if rank == 0:
                        self.model.save(self._save_location, self._nn)
                        self._save_optimizer_state(optimizer)
                else:
                    bad_epochs += 1
                    if bad_epochs >= patience:
                        LOGGER.warning("After %s epochs loss has not improved. Stopping.", bad_epochs)
                        break
        finally:
            self._drop_prestarted_pool()
            if prev_gemm_mode is not None:
                hip_ops.set_msg_gemm_mode(prev_gemm_mode)
        return best
