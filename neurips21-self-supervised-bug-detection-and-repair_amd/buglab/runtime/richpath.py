"""Local-filesystem stand-in for dpu_utils.utils.RichPath (reference buglab/models/train.py:27,74,144;
buglab/utils/msgpackutils.py:30,39): only `create`, `iterate_filtered_files_in_dir`,
`to_local_path().path`, `exists`, `join`.  Azure blob paths are out of scope (no network)."""
import glob
import os
from typing import Iterator, Optional


class RichPath:
    def __init__(self, path: str):
        self.path = str(path)

    @classmethod
    def create(cls, path: str, azure_info_path: Optional[str] = None) -> "RichPath":
        if str(path).startswith("azure://"):
            raise NotImplementedError("Azure storage paths are not supported in this build (no network)")
        return cls(path)

    def iterate_filtered_files_in_dir(self, file_pattern: str) -> Iterator["RichPath"]:
        if os.path.isfile(self.path):
            yield self
            return
        for p in sorted(glob.glob(os.path.join(self.path, file_pattern))):
            yield RichPath(p)

    def to_local_path(self) -> "RichPath":
        return self

    def exists(self) -> bool:
        return os.path.exists(self.path)

    def join(self, name: str) -> "RichPath":
        return RichPath(os.path.join(self.path, name))

    def __lt__(self, other):
        return self.path < other.path

    def __repr__(self):
        return f"RichPath({self.path!r})"


def run_and_debug(fn, enable_debugging: bool = False):
    try:
        return fn()
    except Exception:
        if enable_debugging:
            import pdb
            import traceback

            traceback.print_exc()
            pdb.post_mortem()
        raise
