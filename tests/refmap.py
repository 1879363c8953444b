"""Name/layout mapping between the reference's GnnBugLabModule state_dict (nn.Linear weights are
[out, in]) and this repo's parameter names ([in, out] weights)."""
import numpy as np
import torch

_LOC = "_GnnBugLabModule__localization_module."
_TXT = "_text_repair_module._TextRepairModule__text_rewrite_"
_VAR = "_varmisuse_module._SingleCandidateNodeSelectorModule__candidate_scorer._layers."
_SWP = "_argswap_module._CandidatePairSelectorModule__pair_scorer._layers."

# ours -> (reference name, transform)
MAP = {
    "loc.Ws": (_LOC + "_summary_repr.weight", "T"),
    "loc.bs": (_LOC + "_summary_repr.bias", None),
    "loc.W1": (_LOC + "_l1.weight", "T"),
    "loc.b1": (_LOC + "_l1.bias", None),
    "loc.w": (_LOC + "_repr_to_localization_score.weight", "row0"),
    "text.emb": (_TXT + "embeddings.weight", None),
    "text.W1": (_TXT + "scorer._layers.0.weight", "T"),
    "text.b1": (_TXT + "scorer._layers.0.bias", None),
    "text.w2": (_TXT + "scorer._layers.2.weight", "row0"),
    "text.b2": (_TXT + "scorer._layers.2.bias", None),
    "var.W1": (_VAR + "0.weight", "T"),
    "var.b1": (_VAR + "0.bias", None),
    "var.w2": (_VAR + "2.weight", "row0"),
    "var.b2": (_VAR + "2.bias", None),
    "swap.W1": (_SWP + "0.weight", "T"),
    "swap.b1": (_SWP + "0.bias", None),
    "swap.w2": (_SWP + "2.weight", "row0"),
    "swap.b2": (_SWP + "2.bias", None),
}


def _tx(a, how):
    if how == "T":
        return np.ascontiguousarray(a.T)
    if how == "row0":
        return np.ascontiguousarray(a[0])
    return a


def head_params_from_golden(z, prefix="w_", dtype=torch.float32):
    return {ours: torch.from_numpy(_tx(z[prefix + ref], how)).to(dtype) for ours, (ref, how) in MAP.items()}


def golden_minibatch(z):
    """Rebuild the minibatch dict (numpy) the golden case was generated with."""
    refs = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    mb = {k[3:]: z[k] for k in z.files if k.startswith("mb_")}
    mb["graph_data"] = {
        "reference_node_ids": refs,
        "reference_node_graph_idx": {"candidate_nodes": z["refg_candidate_nodes"]},
        "num_graphs": int(z["B"]),
    }
    return mb
