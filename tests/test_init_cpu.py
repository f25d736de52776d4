"""SURVEY.md §8 row a-13 — the reference's initialisers, checked statistically on the product modules (CPU: construction
never touches the GPU).

  init_params                 mDT/src/modules/graphormer_layers.py:7-13   Linear N(0, 0.02/sqrt(n_layers)), bias 0;
                              Embedding N(0, 0.02) INCLUDING the padding_idx row (nn.Embedding zeroed it, the
                              apply() that follows overwrites it)
  reset_parameters            mDT/src/modules/multihead_attention.py:75-89  xavier_uniform gain 1/sqrt(2) on q, k, v;
                              xavier_uniform out_proj, bias 0
  init_graphormer_params      mDT/src/modules/multigraphormer_graph_encoder.py:18-39 (--apply-graphormer-init)
                              Linear / Embedding / q,k,v N(0, 0.02), Linear bias 0, padding_idx row zero
"""
import math

import torch
import torch.nn as nn


def _std_ok(t, std, tol=0.06):
    s = float(t.float().std())
    return abs(s - std) <= tol * std and abs(float(t.float().mean())) < 4 * std / math.sqrt(t.numel())


def test_init_params_embeddings_and_padding_rows():
    from multimodaldiscussiontransformer_amd.modules import GraphAttnBias, GraphNodeFeature
    torch.manual_seed(0)
    gnf = GraphNodeFeature(num_heads=8, num_atoms=64, num_in_degree=512, num_out_degree=512, hidden_dim=256, n_layers=6)
    for emb in (gnf.atom_encoder, gnf.in_degree_encoder, gnf.out_degree_encoder, gnf.graph_token):
        assert _std_ok(emb.weight.data, 0.02), float(emb.weight.std())
    for emb in (gnf.atom_encoder, gnf.in_degree_encoder, gnf.out_degree_encoder):
        assert emb.padding_idx == 0
        row = emb.weight.data[0]
        assert float(row.abs().max()) > 0.0 and _std_ok(row, 0.02, tol=0.25), "padding_idx row must be OVERWRITTEN by init_params"
    gab = GraphAttnBias(num_heads=8, num_atoms=64, num_edges=96, num_spatial=512, num_edge_dis=16, hidden_dim=256,
                        edge_type="multi_hop", multi_hop_max_dist=5, n_layers=6)
    assert _std_ok(gab.spatial_pos_encoder.weight.data, 0.02)
    assert float(gab.spatial_pos_encoder.weight.data[0].abs().max()) > 0.0
    assert _std_ok(gab.edge_encoder.weight.data, 0.02, tol=0.15)
    assert gab.graph_token_virtual_distance.weight.shape == (1, 8)


def test_init_params_linear_std_scales_with_layers():
    from multimodaldiscussiontransformer_amd.modules.graphormer_layers import init_params
    torch.manual_seed(1)
    for n_layers in (1, 4, 12):
        lin = nn.Linear(512, 512)
        lin.bias.data.fill_(3.0)
        init_params(lin, n_layers)
        assert _std_ok(lin.weight.data, 0.02 / math.sqrt(n_layers))
        assert float(lin.bias.abs().max()) == 0.0


def test_multihead_attention_reset_parameters():
    from multimodaldiscussiontransformer_amd.modules import MultiheadAttention
    torch.manual_seed(2)
    D = 768
    mha = MultiheadAttention(D, 12, dropout=0.0, self_attention=True)
    a_qkv = (1 / math.sqrt(2)) * math.sqrt(6.0 / (D + D))        # xavier_uniform bound with gain 1/sqrt(2)
    a_out = math.sqrt(6.0 / (D + D))
    for i, proj in enumerate((mha.q_proj, mha.k_proj, mha.v_proj)):
        w = proj.weight.data
        assert w.shape == (D, D)
        assert float(w.abs().max()) <= a_qkv + 1e-7 and float(w.abs().max()) > 0.98 * a_qkv
        assert _std_ok(w, a_qkv / math.sqrt(3.0), tol=0.02)
    # the three projections are drawn independently (one fused tensor here, three nn.Linear in the reference)
    assert float((mha.q_proj.weight.data - mha.k_proj.weight.data).abs().max()) > 0.0
    w = mha.out_proj.weight.data
    assert float(w.abs().max()) <= a_out + 1e-7 and float(w.abs().max()) > 0.98 * a_out
    assert _std_ok(w, a_out / math.sqrt(3.0), tol=0.02)
    assert float(mha.out_proj.bias.data.abs().max()) == 0.0
    bound = 1 / math.sqrt(D)                                       # nn.Linear's default bias init is kept for q, k, v
    assert float(mha.qkv_bias.data.abs().max()) <= bound


def test_init_graphormer_params():
    from multimodaldiscussiontransformer_amd.modules import MultiheadAttention, init_graphormer_params
    torch.manual_seed(3)
    lin = nn.Linear(256, 384)
    lin.bias.data.fill_(1.0)
    emb = nn.Embedding(300, 128, padding_idx=0)
    emb_np = nn.Embedding(300, 128)
    mha = MultiheadAttention(256, 8, dropout=0.0, self_attention=True)
    out_before = mha.out_proj.weight.data.clone()
    for m in (lin, emb, emb_np, mha):
        m.apply(init_graphormer_params)
    assert _std_ok(lin.weight.data, 0.02) and float(lin.bias.abs().max()) == 0.0
    assert _std_ok(emb.weight.data[1:], 0.02) and float(emb.weight.data[0].abs().max()) == 0.0      # padding row zeroed HERE
    assert _std_ok(emb_np.weight.data, 0.02) and float(emb_np.weight.data[0].abs().max()) > 0.0
    for proj in (mha.q_proj, mha.k_proj, mha.v_proj):
        assert _std_ok(proj.weight.data, 0.02)
    # out_proj is an nn.Linear child: re-drawn N(0, 0.02) by the same apply()
    assert _std_ok(mha.out_proj.weight.data, 0.02) and not torch.equal(out_before, mha.out_proj.weight.data)


def test_model_level_flag_applies_graphormer_init():
    from types import SimpleNamespace
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    tiny = dict(dim=128, layers=4, heads=4, intermediate=128)
    args = SimpleNamespace(num_bottleneck_tokens=2, num_fusion_layers=1, encoder_embed_dim=128, encoder_ffn_embed_dim=128,
                           encoder_attention_heads=4, apply_graphormer_init=True, dropout=0.0, attention_dropout=0.0, act_dropout=0.0,
                           bert_config=dict(tiny, vocab=512, max_pos=32, type_vocab=2), vit_config=dict(tiny, image_size=32, patch=16))
    torch.manual_seed(4)
    model = GraphormerModel.build_model(args, task=None)
    ge = model.encoder.graph_encoder
    assert float(ge.graph_node_feature.in_degree_encoder.weight.data[0].abs().max()) == 0.0
    assert _std_ok(ge.layers[0].layers[0].self_attn.q_proj.weight.data, 0.02, tol=0.1)
    assert _std_ok(ge.layers[0].layers[0].fc1.weight.data, 0.02, tol=0.1)
    args.apply_graphormer_init = False
    model = GraphormerModel.build_model(args, task=None)
    ge = model.encoder.graph_encoder
    assert float(ge.graph_node_feature.in_degree_encoder.weight.data[0].abs().max()) > 0.0       # init_params overwrote it
