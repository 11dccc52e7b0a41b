"""Quantised GCN layers over the QGTC operators — the working counterpart of the reference's
QGTC_conv.py (which is dead code there: it calls `QGTC.bitMM2Bit` with 4 arguments and
`QGTC.val2bit` with 2 against 8- and 4-argument bindings, QGTC_conv.py:15-21,57-64, and
`GCNConv_Qnt.__init__` calls `super(GCNConv, self)`, :40). Same class names and the same two-step
aggregation (QGTC_conv.py:14-22: X·W, then A·(XW)), with the operand layouts the kernels actually
need: X·W is re-packed in the cols layout by `bitMM2Bit_col` so that it can be the right operand of
A·(XW) (what unitest.py:100-109 does).

Inference only, like the reference (its backward is `pass`, QGTC_conv.py:24-27).
"""
from __future__ import annotations

import torch

import QGTC  # the HIP extension; there is no fallback


class Aggregation_Qnt(torch.autograd.Function):
    """One quantised GCN layer on packed operands: requant(A · requant(X · W)).

    bit_A : rows layout, 1 bit, [n, n]            bit_X : rows layout, act_bit planes, [n, f_in]
    bit_W : cols layout, w_bit planes, [f_in, f_out]
    `output=False` returns the packed activations (rows layout, act_bit planes, [n, f_out]);
    `output=True` returns float32 [n, f_out] (QGTC_conv.py:19-22)."""

    @staticmethod
    def forward(ctx, bit_A, bit_X, bit_W, n, f_in, f_out, act_bit, w_bit, output=False):
        # one extension call per layer (the library's fused-layer entry, qgtc_gcn_layer_batched): X.W re-packed in the
        # cols layout, then A.(XW); word for word what bitMM2Bit_col followed by bitMM2Bit / bitMM2Int returns
        return QGTC.gcn_layer(bit_A, bit_X, bit_W, n, f_in, f_out, 1, act_bit, w_bit, output)

    @staticmethod
    def backward(ctx, d_output):  # the reference has no training path (QGTC_conv.py:24-27)
        raise NotImplementedError("QGTC layers are inference-only")


class GCNConv_Qnt(torch.nn.Module):
    """Two-layer quantised GCN (QGTC_conv.py:38-78): out = A · q(q(A · q(X·W_in)) · W_out)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers=2, w_bit=2, act_bit=3):
        super().__init__()
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.W_in = torch.nn.Parameter(torch.randn(input_dim, hidden_dim))
        self.W_out = torch.nn.Parameter(torch.randn(hidden_dim, output_dim))
        self.w_bit = w_bit
        self.act_bit = act_bit
        # edge-list adjacencies: True checks the indices on the device and reads the flag back (one host sync per
        # forward); callers that build their own induced edge lists (sampler.ClusterIter) can switch it off
        self.validate_edges = True
        self.bit_W_in = None
        self.bit_W_out = None
        self._packed_from = None   # (device, version of W_in, version of W_out) the packed weights were made from

    def _weights_key(self):
        return (self.W_in.device, self.W_in._version, self.W_out._version, self.w_bit)

    def weight_Qnt(self):
        """Pack the weights (cols layout: they are right operands). forward() calls this again whenever the
        parameters moved to another device or were modified in place (optimizer step, load_state_dict)."""
        self.bit_W_in = QGTC.val2bit(self.W_in.detach().contiguous(), self.w_bit, True, False)
        self.bit_W_out = QGTC.val2bit(self.W_out.detach().contiguous(), self.w_bit, True, False)
        self._packed_from = self._weights_key()

    def A_Qnt(self, A):
        """A: dense float [n, n], or a (src, dst, n) edge list (packed without the dense detour)."""
        if isinstance(A, (tuple, list)):
            src, dst, n = A
            return QGTC.pack_edges(src, dst, n, n, 1, self.validate_edges)
        return QGTC.val2bit(A.contiguous(), 1, False, False)

    def X_Qnt(self, X):
        return QGTC.val2bit(X.contiguous(), self.act_bit, False, False)

    def forward(self, A, X):
        """X: node embeddings [n_nodes, n_dim]; A: the subgraph's adjacency (dense or edge list)."""
        if self.bit_W_in is None or self._packed_from != self._weights_key():
            self.weight_Qnt()
        assert X.device == self.W_in.device, "inputs and weights must be on the same device"
        n = X.size(0)
        bit_A = self.A_Qnt(A)
        bit_X = self.X_Qnt(X)
        bit_h = Aggregation_Qnt.apply(bit_A, bit_X, self.bit_W_in, n, self.input_dim, self.hidden_dim,
                                      self.act_bit, self.w_bit, False)
        return Aggregation_Qnt.apply(bit_A, bit_h, self.bit_W_out, n, self.hidden_dim, self.output_dim,
                                     self.act_bit, self.w_bit, True)


class GCNConv(torch.nn.Module):
    """The float reference layer pair of QGTC_conv.py:101-121 (A · ((A · (X·W_in)) · W_out))."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers=2):
        super().__init__()
        self.W_in = torch.nn.Parameter(torch.randn(input_dim, hidden_dim))
        self.W_out = torch.nn.Parameter(torch.randn(hidden_dim, output_dim))

    def forward(self, A, X):
        return torch.mm(A, torch.mm(torch.mm(A, torch.mm(X, self.W_in)), self.W_out))
