#!/usr/bin/env python
"""A `*.msgpack.l.gz` shard written by the REFERENCE's own `save_msgpack_l_gz` (buglab/utils/msgpackutils.py:17-21), for the
on-disk format row:

    python tests/golden/make_golden_shard.py     # rewrites tests/golden/reference_shard.msgpack.l.gz (+ .json)

The datapoints are synthetic BugLab graphs (`make_buglab_dataset`) plus the corner cases a reader has to survive: a `None`
element (the reference's loader skips it), non-ASCII identifiers, empty edge lists.  The JSON next to it is what the
reference's own `load_msgpack_l_gz` reads back."""
import json
import os
import sys

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, OUT)
import make_golden as MG  # noqa: E402


def main():
    MG._install_stubs()
    sys.path.insert(0, os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd"))
    from buglab.data.synthetic import make_buglab_dataset  # the generator only; everything below is the reference's code

    data = make_buglab_dataset(5, seed=4)
    for k in list(sys.modules):
        if k == "buglab" or k.startswith("buglab."):
            del sys.modules[k]
    sys.path.insert(0, "/root/reference")
    from buglab.utils.msgpackutils import load_msgpack_l_gz, save_msgpack_l_gz  # noqa: reference code

    data[1]["graph"]["nodes"][0] = "größe_Ünïcode"
    data[2]["graph"]["edges"]["NextToken"] = []
    elements = [data[0], None, data[1], data[2], data[3], data[4]]
    path = os.path.join(OUT, "reference_shard.msgpack.l.gz")
    save_msgpack_l_gz(elements, path)
    back = list(load_msgpack_l_gz(path))
    with open(os.path.join(OUT, "reference_shard.json"), "w") as f:
        json.dump(back, f)
    print(f"wrote {len(elements)} elements ({os.path.getsize(path)} bytes); read back {len(back)}")


if __name__ == "__main__":
    main()
