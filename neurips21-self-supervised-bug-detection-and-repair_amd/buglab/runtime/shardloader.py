"""Multi-process shard loading: worker processes read `*.msgpack.l.gz` shards (native reader), tensorise the
datapoints and hand the tensorised samples to the trainer, which only collates (native collator) and drives the GPU.

Why: reading + tensorising on the fly is Python-bound at ~500-600 graphs/s per process, the device consumes
~3 000 graphs/s (config c2).  The reference parallelises the same stage with threads (`parallelize=True`,
ptgnn) or ptgnn's `use_multiprocessing`; here the unit of parallelism is the shard file, so no datapoint is
pickled -- only the (NumPy) tensorised samples cross the process boundary.

Host-only; nothing here touches the HIP extension (workers never initialise a GPU)."""
import multiprocessing as mp
import os
import queue as queue_mod
import random
from typing import Iterable, Iterator, List, Optional, Sequence

from buglab.runtime.richpath import RichPath
from buglab.utils.msgpackutils import load_all_msgpack_l_gz, load_msgpack_l_gz

_DONE = "__shard_worker_done__"


class ShardDataset:
    """The datapoints of a directory of shards -- iterable like the reference's `LazyDataIterable(load_all_msgpack_l_gz...)`
    (train.py:76-91), and additionally aware of its files so that loading can be spread over processes."""

    def __init__(self, path, shuffle: bool = False, take_only_first_n_files: Optional[int] = None,
                 limit_num_yielded_elements: Optional[int] = None, seed: int = 0):
        self.path = path if isinstance(path, RichPath) else RichPath.create(str(path))
        self.shuffle, self.take_only_first_n_files = shuffle, take_only_first_n_files
        self.limit_num_yielded_elements = limit_num_yielded_elements
        self.seed, self.epoch = seed, 0

    def set_epoch(self, epoch: int) -> None:
        """The file order of an epoch is a function of (seed, epoch) only: every rank of a data-parallel run sees the
        SAME order, so that the per-rank shares of the stream are disjoint and cover it."""
        self.epoch = int(epoch)

    def shard_files(self) -> List[str]:
        files = sorted(self.path.iterate_filtered_files_in_dir("*.msgpack.l.gz"))
        if self.take_only_first_n_files is not None:
            files = files[: self.take_only_first_n_files]
        files = [f.to_local_path().path for f in files]
        if self.shuffle:
            random.Random(self.seed * 1_000_003 + self.epoch).shuffle(files)
        return files

    def __iter__(self):
        n = 0
        for f in self.shard_files():  # same (seeded) order as the loader processes use
            for d in _read_shard(f):  # an unreadable shard is reported and skipped (reference msgpackutils.py:45-46)
                if d is None:
                    continue
                yield d
                n += 1
                if self.limit_num_yielded_elements is not None and n > self.limit_num_yielded_elements:
                    return  # same cut-off as load_all_msgpack_l_gz (reference msgpackutils.py:40-41)


def default_num_workers() -> int:
    """BUGLAB_LOADER_WORKERS (0 = load in the trainer process); default: up to 8, leaving cores for the ranks of a node."""
    env = os.environ.get("BUGLAB_LOADER_WORKERS")
    if env is not None:
        return max(0, int(env))
    cores = os.cpu_count() or 1
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return max(0, min(8, cores // max(1, world) - 1))


def _read_shard(f: str) -> Iterator:
    """Datapoints of one shard; a shard that cannot be read is reported and skipped like the reference does
    (msgpackutils.py:45-46).  ONLY reading is guarded: tensorize / collate errors are bugs or bad samples and travel to
    the consumer (`_WorkerError`) instead of silently dropping the rest of the shard."""
    it = iter(load_msgpack_l_gz(f))
    while True:
        try:
            d = next(it)
        except StopIteration:
            return
        except Exception as e:
            print(f"Error loading {f}: {e}.")
            return
        yield d


class _WorkerError:
    def __init__(self, where: str, exc: BaseException):
        import traceback

        self.text = f"loader process failed in {where}: {exc!r}\n{traceback.format_exc()}"


def _rank_stream(files: Sequence[str], rank: int, world: int, stop) -> Iterator:
    """This rank's datapoints of the worker's files: per file, windows split by message count (see
    buglab.runtime.distributed.balanced_rank_share) -- every rank's worker w reads the same files in the same order."""
    from buglab.runtime.distributed import balanced_rank_share

    for f in files:
        for d in balanced_rank_share(_read_shard(f), rank, world):
            if stop.is_set():
                return
            yield d


def _worker(model, files: Sequence[str], rank: int, world: int, out_q, stop) -> None:
    try:
        for d in _rank_stream(files, rank, world, stop):
            t = model.tensorize(d)
            if t is not None:  # dropped sample (reference gnn.py:404-405)
                out_q.put(t)
    except BaseException as e:
        out_q.put(_WorkerError("tensorize", e))
    finally:
        out_q.put(_DONE)


def _minibatch_worker(model, files: Sequence[str], rank: int, world: int, max_minibatch_size: int, packed: bool, out_q, stop) -> None:
    try:
        mb, n = model.initialize_minibatch(), 0
        for d in _rank_stream(files, rank, world, stop):
            t = model.tensorize(d)
            if t is None:
                continue
            keep = model.extend_minibatch_with(t, mb)
            n += 1
            if not keep or n >= max_minibatch_size:
                out_q.put(_ship(model.collate_minibatch(mb), packed))
                mb, n = model.initialize_minibatch(), 0
        if n > 0 and not stop.is_set():
            out_q.put(_ship(model.collate_minibatch(mb), packed))
    except BaseException as e:
        out_q.put(_WorkerError("tensorize / collate", e))
    finally:
        out_q.put(_DONE)


def _ship(mb_np, packed: bool):
    """What a loader process puts on the queue for one minibatch.  packed: the int32 blob of `pack_minibatch` goes through a
    POSIX shared-memory segment (written here, mapped + unlinked by the consumer) and only its name and the small metadata
    are pickled -- a c2 minibatch is ~7.5 MB, and reading + unpickling it from the queue's pipe in the trainer process cost as
    much as a whole device step."""
    if not packed:
        return mb_np
    from multiprocessing import shared_memory

    import numpy as np

    from buglab.data.collate import pack_minibatch, packed_size

    shm = shared_memory.SharedMemory(create=True, size=4 * packed_size(mb_np))
    try:
        _, meta = pack_minibatch(mb_np, out=np.ndarray((shm.size // 4,), dtype=np.int32, buffer=shm.buf))
        return ("__packed__", shm.name, meta)
    finally:
        shm.close()  # the segment lives on until the consumer unlinks it


def receive_packed(item, device):
    """Consumer side of `_ship(..., packed=True)`: map the segment, stage + upload it, unlink it."""
    from multiprocessing import shared_memory

    import numpy as np

    from buglab.data.collate import upload_packed

    _, name, meta = item
    shm = shared_memory.SharedMemory(name=name)
    try:
        return upload_packed(np.ndarray((int(meta["total"]),), dtype=np.int32, buffer=shm.buf), meta, device)
    finally:
        shm.close()
        shm.unlink()


def collated_minibatches_parallel(model, files: Sequence[str], num_workers: int, max_minibatch_size: int, rank: int = 0,
                                  world: int = 1, packed: bool = False, pool: "Optional[WorkerPool]" = None) -> Iterator:
    """Collated minibatches of `files`: every worker process reads its shard files, tensorises and collates; the
    consumer only copies a minibatch to the device.  packed = False: NumPy dicts (`buglab.data.collate.to_device` them);
    packed = True: items for `receive_packed` (the int32 blob travels through shared memory, not the queue's pipe).
    pool: a `minibatch_pool(...)` of the same arguments that was started earlier (its start-up then overlapped other work)."""
    if pool is None:
        pool = minibatch_pool(model, files, num_workers, max_minibatch_size, rank, world, packed)
    yield from pool


def minibatch_pool(model, files: Sequence[str], num_workers: int, max_minibatch_size: int, rank: int = 0, world: int = 1,
                   packed: bool = False) -> "WorkerPool":
    """The loader processes behind `collated_minibatches_parallel`, not started yet: `.start()` forks them (e.g. while the
    previous epoch's validation is still running), iterating yields their minibatches, `.close()` shuts them down."""
    return WorkerPool(_minibatch_worker, model, files, num_workers, (rank, world, max_minibatch_size, packed), 8, None)


def tensorize_shards_parallel(model, files: Sequence[str], num_workers: int, rank: int = 0, world: int = 1,
                              limit_num_yielded_elements: Optional[int] = None) -> Iterator:
    """Tensorised samples of `files`, produced by `num_workers` forked processes (worker w takes files w, w + W, ...).
    Under data parallelism every rank keeps the datapoints with index % world == rank of each file, like the in-process
    loader.  Sample order across files is not deterministic (files are shuffled by the trainer anyway)."""
    yield from WorkerPool(_worker, model, files, num_workers, (rank, world), 512, limit_num_yielded_elements)


def _discard(item) -> None:
    if isinstance(item, tuple) and len(item) == 3 and item[0] == "__packed__":
        from multiprocessing import shared_memory

        try:
            shm = shared_memory.SharedMemory(name=item[1])
            shm.close()
            shm.unlink()
        except FileNotFoundError:
            pass


class WorkerPool:
    """`num_workers` forked loader processes (worker w takes files w, w + W, ...) feeding one queue.  Iterate to consume;
    `start()` may be called ahead of time so that forking and the first shards overlap other work; `close()` is idempotent."""

    def __init__(self, target, model, files: Sequence[str], num_workers: int, extra_args, queue_size: int, limit: Optional[int]):
        self.files = list(files)
        self.num_workers = max(1, min(num_workers, len(self.files))) if self.files else 0
        self._args = (target, model, extra_args, queue_size)
        self.limit = limit
        self._procs = None
        self._q = self._stop = None
        self._closed = False
        self._finished = False  # every worker delivered its end marker (nothing can be left in the queue)

    def start(self) -> "WorkerPool":
        if self._procs is not None or self.num_workers == 0 or self._closed:
            return self
        target, model, extra_args, queue_size = self._args
        # One resource tracker for the consumer and all workers: started BEFORE the fork so that the children inherit it.
        # (A worker forked earlier would start its own tracker, which "cleans up" -- unlinks -- the shared-memory segments
        # the worker created as soon as the worker exits, i.e. while its last minibatches are still waiting in the queue.)
        from multiprocessing import resource_tracker

        resource_tracker.ensure_running()
        ctx = mp.get_context("fork")  # the model (vocabulary, caches) is shared copy-on-write; nothing is pickled to start
        self._q = ctx.Queue(maxsize=queue_size)
        self._stop = ctx.Event()
        procs = [ctx.Process(target=target, args=(model, self.files[w::self.num_workers], *extra_args, self._q, self._stop), daemon=True)
                 for w in range(self.num_workers)]
        # The children inherit the interpreter's heap, including any unreachable-but-not-yet-collected objects of the
        # trainer (device tensors, HIP events / streams in reference cycles).  A collection inside a child would run their
        # destructors there, and HIP does not survive a fork: collect here, then freeze what exists so that no child's
        # collector ever looks at it; the parent thaws its own heap again once the children are running.
        import gc

        gc.collect()
        gc.freeze()
        try:
            for p in procs:
                p.start()
        finally:
            gc.unfreeze()
        self._procs = procs
        return self

    def __iter__(self) -> Iterator:
        if self.num_workers == 0:
            return
        self.start()
        procs, out_q = self._procs, self._q
        done, n = 0, 0
        try:
            while done < self.num_workers:
                try:
                    item = out_q.get(timeout=1.0)
                except queue_mod.Empty:
                    if not any(p.is_alive() for p in procs) and out_q.empty():
                        raise RuntimeError(f"{self.num_workers - done} loader process(es) exited without finishing their shards "
                                           "(killed? out of memory?): the epoch would silently be incomplete")
                    continue
                if isinstance(item, str) and item == _DONE:
                    done += 1
                    self._finished = done == self.num_workers
                    continue
                if isinstance(item, _WorkerError):
                    raise RuntimeError(item.text)
                yield item
                n += 1
                if self.limit is not None and n >= self.limit:
                    break
        finally:
            self.close()

    def close(self) -> None:
        """Shutdown.  Workers see `stop`, finish the item they are putting and exit; the queue is drained meanwhile so that
        nobody stays blocked on a full pipe (shared-memory segments of dropped items are unlinked).  A worker that is
        still alive after the grace period is killed -- and from then on the pipe may end in a truncated message, on
        which a read would block for ever: no reads after a kill."""
        if self._closed:
            return
        self._closed = True
        if self._procs is None:
            return
        import time as _time

        procs, out_q = self._procs, self._q
        self._stop.set()
        if self._finished:
            # The normal end of an epoch: the queue is empty and every worker is on its way out.  Reaping 32 exiting
            # processes took 0.1-0.25 s of every epoch on the trainer's thread (seconds on a loaded host): a daemon thread
            # does it instead (stragglers are killed there; daemonic children die with the interpreter in any case).
            import threading

            def reap():
                for p in procs:
                    p.join(timeout=5.0)
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                        p.join(timeout=2.0)
                out_q.close()
                out_q.cancel_join_thread()

            threading.Thread(target=reap, name="buglab-loader-reaper", daemon=True).start()
            return
        deadline = _time.monotonic() + 5.0
        while any(p.is_alive() for p in procs) and _time.monotonic() < deadline:
            try:
                _discard(out_q.get(timeout=0.05))
            except queue_mod.Empty:
                pass
        stuck = [p for p in procs if p.is_alive()]
        for p in stuck:
            p.terminate()
        for p in procs:
            p.join(timeout=2.0)
        if not stuck:
            try:  # every writer exited on its own: what is left in the pipe are whole messages
                while True:
                    _discard(out_q.get(timeout=0.05))
            except queue_mod.Empty:
                pass
        out_q.close()
        out_q.cancel_join_thread()
