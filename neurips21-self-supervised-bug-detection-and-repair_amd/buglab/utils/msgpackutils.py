"""`*.msgpack.l.gz` streaming IO -- same functions as reference buglab/utils/msgpackutils.py
(load_msgpack_l_gz :11-14, save_msgpack_l_gz :17-21, load_all_msgpack_l_gz :24-46)."""
import gzip
import random
from collections import OrderedDict
from os import PathLike
from typing import Any, Iterable, Iterator, Optional

import msgpack

from buglab.runtime.richpath import RichPath


def _use_native_reader() -> bool:
    """The C++ reader (buglab/data/native.py) is used when its library is built, unless BUGLAB_NATIVE_READER=0."""
    import os

    if os.environ.get("BUGLAB_NATIVE_READER", "1") == "0":
        return False
    from buglab.data import native

    return native.available()


def load_msgpack_l_gz(filename: PathLike, native: Optional[bool] = None) -> Iterator[Any]:
    if native if native is not None else _use_native_reader():
        from buglab.data.native import load_msgpack_l_gz_native

        yield from load_msgpack_l_gz_native(filename)
        return
    with gzip.open(filename) as f:
        unpacker = msgpack.Unpacker(f, raw=False, object_pairs_hook=OrderedDict, strict_map_key=False)
        yield from unpacker


def save_msgpack_l_gz(data: Iterable[Any], filename: PathLike) -> None:
    with gzip.GzipFile(filename, "wb") as out_file:
        packer = msgpack.Packer(use_bin_type=True)
        for element in data:
            out_file.write(packer.pack(element))


def load_all_msgpack_l_gz(path, shuffle: bool = False, take_only_first_n_files: Optional[int] = None,
                          limit_num_yielded_elements: Optional[int] = None) -> Iterator:
    if not isinstance(path, RichPath):
        path = RichPath.create(str(path))
    all_files = sorted(path.iterate_filtered_files_in_dir("*.msgpack.l.gz"))
    if take_only_first_n_files is not None:
        all_files = all_files[:take_only_first_n_files]
    if shuffle:
        random.shuffle(all_files)
    sample_idx = 0
    for msgpack_file in all_files:
        try:
            for element in load_msgpack_l_gz(msgpack_file.to_local_path().path):
                if element is not None:
                    sample_idx += 1
                    yield element
                if limit_num_yielded_elements is not None and sample_idx > limit_num_yielded_elements:
                    return
        except Exception as e:  # reference :45-46: data-file errors are printed and skipped
            print(f"Error loading {msgpack_file}: {e}.")
