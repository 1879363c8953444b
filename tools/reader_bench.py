#!/usr/bin/env python
"""Host-side decode + tensorise throughput of the Python reader vs the native reader (CPU only).
    python tools/reader_bench.py [--graphs 200] [--nodes 1500]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
import numpy as np

from buglab.data.synthetic import make_buglab_datapoint
from buglab.models.modelregistry import load_model
from buglab.utils.msgpackutils import load_msgpack_l_gz, save_msgpack_l_gz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=200)
    ap.add_argument("--nodes", type=int, default=1500)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    data = [make_buglab_datapoint(rng, num_syntax_nodes=a.nodes, num_tokens=a.nodes // 2, buggy=bool(i % 2)) for i in range(a.graphs)]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "shard.msgpack.l.gz")
        save_msgpack_l_gz(data, path)
        from pathlib import Path

        model, _, _ = load_model({"modelName": "gnn-mlp"}, Path(d) / "m.pkl.gz")
        for x in load_msgpack_l_gz(path, native=False):
            model.update_metadata_from(x)
        model.finalize_metadata()
        print(f"shard: {a.graphs} graphs, ~{a.nodes * 3 // 2} nodes each, {os.path.getsize(path) / 1e6:.1f} MB gz")
        for native in (False, True):
            for what in ("decode", "decode+tensorize"):
                t0 = time.perf_counter()
                n = 0
                for x in load_msgpack_l_gz(path, native=native):
                    if what != "decode":
                        model.tensorize(x)
                    n += 1
                dt = time.perf_counter() - t0
                print(f"{'native' if native else 'python'} {what:17s}: {n / dt:8.1f} graphs/s (one core)")


if __name__ == "__main__":
    main()
