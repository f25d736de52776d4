"""Drop-in for mDT/src/modules/graphormer_layers.py (``GraphNodeFeature``, ``GraphAttnBias``).

Parameter names, shapes, initialisation and the dead parameters (``atom_encoder``,
``edge_encoder``, ``edge_dis_encoder``) match the reference so checkpoints load unchanged.
Standalone ``forward`` calls reproduce the reference outputs through HIP kernels; inside the
fused encoder neither module materialises anything — node features are built by the scatter
kernel that assembles the graph tokens, and the structural bias is evaluated inside the
attention kernel.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import engine as E
from .. import ops


def init_params(module, n_layers):
    """graphormer_layers.py:7-13 (note: also overwrites the padding_idx rows)."""
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02 / math.sqrt(n_layers))
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=0.02)


class GraphNodeFeature(nn.Module):
    """x + in_degree_embedding + out_degree_embedding, graph token prepended."""

    def __init__(self, num_heads, num_atoms, num_in_degree, num_out_degree, hidden_dim, n_layers):
        super().__init__()
        self.num_heads = num_heads
        self.num_atoms = num_atoms
        self.atom_encoder = nn.Embedding(num_atoms + 1, hidden_dim, padding_idx=0)       # never used (as in the reference)
        self.in_degree_encoder = nn.Embedding(num_in_degree, hidden_dim, padding_idx=0)
        self.out_degree_encoder = nn.Embedding(num_out_degree, hidden_dim, padding_idx=0)
        self.graph_token = nn.Embedding(1, hidden_dim)
        self.apply(lambda module: init_params(module, n_layers=n_layers))

    def forward(self, x, in_degree, out_degree):
        n_graph, n_node, D = x.shape
        T = n_node + 1
        dev = x.device
        ind = in_degree.to(torch.int32).contiguous().view(-1)
        outd = ind if out_degree is in_degree else out_degree.to(torch.int32).contiguous().view(-1)
        rows = torch.arange(n_graph * n_node, dtype=torch.int32, device=dev)

        def scatter_idx(d):
            idx = torch.full((n_graph, T), -1, dtype=torch.int32, device=dev)
            d2 = d.view(n_graph, n_node)
            idx[:, 1:] = torch.where(d2 > 0, d2, torch.full_like(d2, -1))
            return idx.view(-1)

        graph_rows = (torch.arange(n_graph, device=dev)[:, None] * T + 1 + torch.arange(n_node, device=dev)[None]).to(torch.int32).view(-1)
        in_idx = scatter_idx(ind)
        out_idx = in_idx if outd is ind else scatter_idx(outd)

        def run(tape, xv):
            return (E.graph_node_features(tape, xv, rows, ind, outd, self.in_degree_encoder.weight,
                                          self.out_degree_encoder.weight, self.graph_token.weight, n_graph, T, rows,
                                          graph_rows, n_graph * n_node, in_idx, out_idx),)

        (out,) = E.run_tape(run, [x.contiguous().view(n_graph * n_node, D)], list(self.parameters()))
        return out.view(n_graph, T, D)


class _GraphAttnBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, virt, attn_bias, spatial_pos):
        ctx.save_for_backward(spatial_pos)
        ctx.shapes = (table.shape, virt.shape, table.dtype)
        return ops.graph_attn_bias(attn_bias, spatial_pos, table, virt.view(-1))

    @staticmethod
    def backward(ctx, g):
        (sp,) = ctx.saved_tensors
        tshape, vshape, dtype = ctx.shapes
        B, H, S, _ = g.shape
        g = g.contiguous()
        dt = torch.zeros(tshape, dtype=torch.float32, device=g.device)
        dv = torch.zeros(vshape[-1], dtype=torch.float32, device=g.device)
        # rows of the [B*H*S*S] gradient viewed per (b, h, i): scatter the node x node block into the table
        # via the row scatter-add kernel on a [B*N*N, H] rearrangement
        N = S - 1
        blk = g[:, :, 1:, 1:].permute(0, 2, 3, 1).reshape(B * N * N, H).contiguous()
        idx = sp.reshape(-1).to(torch.int32)
        idx = torch.where(idx > 0, idx, torch.full_like(idx, -1))
        ops.row_scatter_add(dt, idx, blk, B * N * N)
        edge = torch.cat([g[:, :, 0, :], g[:, :, 1:, 0]], dim=2).permute(0, 2, 1).reshape(-1, H).contiguous()
        ops.colsum(edge, out=dv)
        return dt.to(dtype), dv.view(vshape).to(dtype), None, None


class GraphAttnBias(nn.Module):
    """Per-head structural attention bias (graphormer_layers.py:53-110)."""

    def __init__(self, num_heads, num_atoms, num_edges, num_spatial, num_edge_dis, hidden_dim, edge_type,
                 multi_hop_max_dist, n_layers):
        super().__init__()
        self.num_heads = num_heads
        self.multi_hop_max_dist = multi_hop_max_dist
        self.edge_encoder = nn.Embedding(num_edges + 1, num_heads, padding_idx=0)          # never used
        self.edge_type = edge_type
        if self.edge_type == "multi_hop":
            self.edge_dis_encoder = nn.Embedding(num_edge_dis * num_heads * num_heads, 1)   # never used
        self.spatial_pos_encoder = nn.Embedding(num_spatial, num_heads, padding_idx=0)
        self.graph_token_virtual_distance = nn.Embedding(1, num_heads)
        self.apply(lambda module: init_params(module, n_layers=n_layers))

    def forward(self, batched_data):
        """→ f32 [n_graph, n_head, T, T]; ``attn_bias`` enters twice (clone :93 + "reset" :108)."""
        attn_bias = batched_data["attn_bias"].float().contiguous()
        spatial_pos = batched_data["spatial_pos"].to(torch.int32).contiguous()
        return _GraphAttnBiasFn.apply(self.spatial_pos_encoder.weight, self.graph_token_virtual_distance.weight,
                                      attn_bias, spatial_pos)
