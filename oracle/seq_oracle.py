"""CPU oracle of the whole `seq-great` / `seq-rat` model forward (reference buglab/models/seqmodel.py:164-396):
token embedding -> + positional table -> LayerNorm -> mask -> relational transformer stack -> the scoring heads and
the loss.  TEST INFRASTRUCTURE ONLY.  It composes pieces that are pinned elsewhere: the transformer block is
`oracle/great_oracle.py` (pinned to the reference's own layers), the heads and the loss are `oracle/buglab_oracle.py`
sections H1-H8 (pinned to the reference's GnnBugLabModule.forward, whose code the sequence module shares line for
line); the subtoken embedder is the graph model's (ptgnn, unpinned).  Dropout off."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from oracle import buglab_oracle as O
from oracle import great_oracle as G


def forward_loss(p: Dict[str, torch.Tensor], mb, cfg: G.GreatConfig, buggy_weight: float = 1.0):
    """p: encoder parameters under the reference's layer names (`layers.{i}....`) plus `embed.table`,
    `positional_encoding` [1, P, D], `input_norm.weight/.bias`, and the head parameters under buglab_oracle's names."""
    gd = mb["graph_data"]
    B, L = int(gd["seq_batch"]), int(gd["seq_len"])
    lens = torch.as_tensor(np.asarray(gd["seq_lens"]), dtype=torch.int64)
    emb = O.embed_nodes(p["embed.table"], gd["token_ids"], gd["token_lens"], 0.0, None).view(B, L, -1)  # seqmodel.py:354-361
    x = emb + p["positional_encoding"][:, :L]  # :369
    x = F.layer_norm(x, (cfg.d_model,), p["input_norm.weight"], p["input_norm.bias"], 1e-5)  # :372 (dropout off)
    valid = torch.arange(L)[None, :] < lens[:, None]
    x = x * valid[:, :, None]  # :375
    # edges back from the query-row CSR: forward entries (even codes) are (sample, source = row, target = key)
    rp, key, code = (np.asarray(gd[k]) for k in ("erow_ptr", "ekey", "ecode"))
    rows = np.repeat(np.arange(B * L), np.diff(rp))
    fwd = code % 2 == 0
    edges = torch.as_tensor(np.stack([rows[fwd] // L, rows[fwd] % L, key[fwd]], 1), dtype=torch.int64).reshape(-1, 3)
    types = torch.as_tensor(code[fwd] // 2, dtype=torch.int64)
    h = G.encoder_stack(p, x, ~valid, edges, types, cfg).reshape(B * L, -1)  # :377-381
    Lng = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int64)
    refs = gd["reference_node_ids"]
    swap_lp, text_lp, var_lp, sel, _ = O.repair_logprobs(p, h, refs, mb["target_rewrites"], mb["rewrite_to_location_group"],
                                                        mb["candidate_symbol_to_location_group"], mb["swapped_pair_to_call_location_group"])
    loc_loss, loc_lp, stats = O.localization_loss(p, h[Lng(refs["candidate_nodes"])], gd["reference_node_graph_idx"]["candidate_nodes"],
                                                  mb["has_bug"], mb["correct_candidate_node_idxs"], buggy_weight)
    repair = -(text_lp[Lng(mb["correct_rewrite_idxs"])].sum() + var_lp[Lng(mb["correct_candidate_symbols"])].sum()
               + swap_lp[Lng(mb["correct_swapped_pair"])].sum()) * buggy_weight
    loss = loc_loss + repair / len(mb["has_bug"])
    return {"loss": loss, "node_reprs": h, "loc_logprobs": loc_lp, "text_logprobs": text_lp, "var_logprobs": var_lp, "swap_logprobs": swap_lp}
