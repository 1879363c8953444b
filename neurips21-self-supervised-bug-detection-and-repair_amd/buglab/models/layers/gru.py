"""Bidirectional GRU layer of `seq-gru` -- MI355X counterpart of one layer of the `torch.nn.GRU(input_size=D, hidden_size=D // 2,
num_layers, bidirectional=True, batch_first=True)` the reference runs over a PackedSequence (buglab/models/seqmodel.py:119-126
construction, :385-392 call).  torch's semantics kept: gate order [r | z | n], b_hn inside the reset product
(n = tanh(W_in x + b_in + r (W_hn h + b_hn))), every sequence processed over its own length, the reverse direction starting at the
last real token, zeros at padded positions, no dropout between layers (the reference passes none), every parameter initialised
uniform(+-1 / sqrt(hidden_size)).

The input projections of both directions and all time steps are ONE row GEMM (hip_ops.gather_linear: MFMA, bias in the epilogue);
the recurrence is csrc/bl_gru_scan.hip (one workgroup per sequence and direction, W_hh in registers).  Parameters are stored
[in, out] with the two directions side by side; `load_torch_gru` / `torch_layout` map from and to torch's names, and
tests/test_seq_gru_gpu.py checks the stack against torch.nn.GRU itself.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from buglab.models import hip_ops


class BiGRULayer(nn.Module):
    def __init__(self, input_size: int, hidden_size: int):
        super().__init__()
        if hidden_size not in (32, 64, 128):
            raise NotImplementedError(f"seq-gru: hidden size per direction {hidden_size} (= hidden_state_size / 2); the recurrence kernel "
                                      "is built for 32, 64 and 128 (hidden_state_size 64 / 128 / 256: csrc/bl_gru_scan.hip)")
        if input_size % 4 != 0:
            raise ValueError("seq-gru: the input width must be a multiple of 4 (row alignment of the GEMM operand)")
        Hh, bound = hidden_size, 1.0 / math.sqrt(hidden_size)
        u = lambda *shape: nn.Parameter(torch.empty(shape).uniform_(-bound, bound))
        self.input_size, self.hidden_size = input_size, Hh
        self.W_ih, self.b_ih = u(input_size, 6 * Hh), u(6 * Hh)   # columns [direction][r | z | n][Hh]
        self.W_hh, self.b_hh = u(2, Hh, 3 * Hh), u(2, 3 * Hh)

    @torch.no_grad()
    def load_torch_gru(self, gru: "nn.GRU", layer: int) -> "BiGRULayer":
        Hh = self.hidden_size
        c = lambda t: t.detach().to(device=self.W_ih.device, dtype=self.W_ih.dtype)
        for d, sfx in enumerate(("", "_reverse")):
            self.W_ih[:, d * 3 * Hh:(d + 1) * 3 * Hh] = c(getattr(gru, f"weight_ih_l{layer}{sfx}").t())
            self.b_ih[d * 3 * Hh:(d + 1) * 3 * Hh] = c(getattr(gru, f"bias_ih_l{layer}{sfx}"))
            self.W_hh[d] = c(getattr(gru, f"weight_hh_l{layer}{sfx}").t())
            self.b_hh[d] = c(getattr(gru, f"bias_hh_l{layer}{sfx}"))
        return self

    def torch_layout(self, layer: int, tensors: Optional[dict] = None) -> dict:
        """this layer's parameters (or {name: tensor} of the same shapes, e.g. gradients) under torch.nn.GRU's names and shapes"""
        Hh = self.hidden_size
        t = tensors if tensors is not None else {k: v.detach() for k, v in self.named_parameters()}
        out = {}
        for d, sfx in enumerate(("", "_reverse")):
            out[f"weight_ih_l{layer}{sfx}"] = t["W_ih"][:, d * 3 * Hh:(d + 1) * 3 * Hh].t()
            out[f"bias_ih_l{layer}{sfx}"] = t["b_ih"][d * 3 * Hh:(d + 1) * 3 * Hh]
            out[f"weight_hh_l{layer}{sfx}"] = t["W_hh"][d].t()
            out[f"bias_hh_l{layer}{sfx}"] = t["b_hh"][d]
        return out

    def forward(self, x: torch.Tensor, lens: torch.Tensor, edges, B: int, L: int, dropout_seed: Optional[int] = None,
                dropout_stream: int = 0, chain: Optional[dict] = None) -> torch.Tensor:
        """x [B * L, input_size] -> [B * L, 2 * hidden_size]; lens int32 [B]; `edges`, dropout arguments: ignored (same call shape as
        the transformer layers)."""
        gi = hip_ops.gather_linear([(x, None)], self.W_ih, self.b_ih)
        return hip_ops.gru_scan(gi, self.W_hh, self.b_hh, lens, B, L)
