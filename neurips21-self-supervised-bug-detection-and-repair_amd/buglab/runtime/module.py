"""`ModuleWithMetrics` -- the ptgnn base class the reference's modules derive from
(reference buglab/models/gnn.py:55,95-114; layers/localizationmodule.py:11,30-52;
layers/fixermodules.py:9,19-29).  Contract pinned by those call sites: subclasses override
`_reset_module_metrics()` and `_module_metrics() -> Dict`; the trainer calls `reset_metrics()` once
per epoch and `report_metrics()` to collect a flat dict over all sub-modules."""
from typing import Any, Dict

from torch import nn


class ModuleWithMetrics(nn.Module):
    def __init__(self):
        super().__init__()

    def _reset_module_metrics(self) -> None:
        pass

    def _module_metrics(self) -> Dict[str, Any]:
        return {}

    def reset_metrics(self) -> None:
        for m in self.modules():
            if isinstance(m, ModuleWithMetrics):
                m._reset_module_metrics()

    def report_metrics(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {}
        for m in self.modules():
            if isinstance(m, ModuleWithMetrics):
                out.update(m._module_metrics())
        return out
