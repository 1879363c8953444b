"""The slice of the third-party `ptgnn` / `dpu_utils` surface that BugLab's `buglab.models` entry
points rely on ("third-party components that ARE the hot path"), written from
scratch for this repo: neither package is installable offline and only the call-site contract the
reference pins is reproduced, not their implementation."""
from buglab.runtime.module import ModuleWithMetrics  # noqa: F401
