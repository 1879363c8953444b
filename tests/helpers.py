"""Shared helpers for the GPU parity tests (tests only; may import the oracle)."""
import numpy as np
import torch

from oracle import buglab_oracle as O

_PREFIX = {
    "edge_embed.": "_gnn.edge_embed.",
    "embed.": "_gnn.embed.",
    "mp.": "_gnn.mp.",
    "loc.": "_localization_module.",
    "text.": "_text_repair_module.",
    "var.": "_varmisuse_module.",
    "swap.": "_argswap_module.",
}


def module_name(oracle_name: str) -> str:
    for k, v in _PREFIX.items():
        if oracle_name.startswith(k):
            return v + oracle_name[len(k):]
    raise KeyError(oracle_name)


def load_oracle_params(module, params):
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[module_name(k)].copy_(v.to(sd[module_name(k)].device))


def module_grads(module):
    named = dict(module.named_parameters())
    inv = {}
    for k in named:
        for o, m in _PREFIX.items():
            if k.startswith(m):
                inv[o + k[len(m):]] = named[k]
    return {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in inv.items()}


def make_case(B=4, n=80, E=400, T=5, H=64, layers=4, vocab=300, C=8, seed=0, degree="uniform", max_degree=512, **kw):
    from buglab.data.collate import collate_samples
    from buglab.data.synthetic import make_samples

    cfg = O.OracleConfig(hidden=H, num_layers=layers, num_edge_types=T, vocab_size=vocab, **kw)
    samples = make_samples(B, seed=seed, num_nodes=n, num_messages=E, num_edge_types=T, vocab_size=vocab,
                           num_candidates=C, degree=degree, max_degree=max_degree)
    mb = collate_samples(samples, T)
    return cfg, samples, mb


def build_module_like(cfg, params=None, device="cuda"):
    from buglab.models.gnn import build_gnn_mlp_module

    m = build_gnn_mlp_module(cfg.hidden, cfg.num_layers, cfg.num_edge_types, cfg.vocab_size, cfg.max_subtokens,
                             cfg.rewrite_vocab_size, cfg.dropout, cfg.msg_act, cfg.buggy_samples_weight, model=cfg.model,
                             edge_feature_size=cfg.edge_feature_size, edge_vocabulary_size=cfg.edge_vocab_size,
                             message_activation_placement=cfg.msg_act_placement,
                             embedder_dropout_placement=cfg.embed_dropout_placement,
                             message_aggregation_function=getattr(cfg, "msg_aggregation", "max")).to(device)
    if params is not None:
        load_oracle_params(m, params)
    return m


def maxdiff(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max())
