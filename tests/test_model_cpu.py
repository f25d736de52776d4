"""CPU checks of the host-side mirror: state-dict naming contract against the oracle's
canonical (reference) parameter table, q/k/v split-merge, layer-count rules, collator parity,
and that the product path refuses to run without a GPU (no silent fallback)."""
import numpy as np
import pytest
import torch

from oracle import cases
from oracle import mdt_ref_cpu as R
from oracle import structure as S
from tests.util_model import ENC, fill_hash_weights, model_args


def build(kind):
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams(kind)
    return hp, GraphormerModel.build_model(model_args(hp), task=None)


@pytest.mark.parametrize("kind", ["A", "B"])
def test_state_dict_matches_reference_names(kind):
    hp, model = build(kind)
    sd = model.state_dict()
    want = R.param_shapes(hp)
    got = {k[len(ENC):]: tuple(v.shape) for k, v in sd.items() if k.startswith(ENC)}
    for alias in ("text_pooler.dense.weight", "vit_pooler.dense.bias"):
        assert alias in got                      # alias keys are emitted like the reference does
    got = {k: v for k, v in got.items() if not k.startswith(("text_pooler.", "vit_pooler."))}
    assert set(got) == set(want), (sorted(set(got) - set(want))[:5], sorted(set(want) - set(got))[:5])
    for k in want:
        assert got[k] == tuple(want[k]), k
    for k in ("encoder.node_encoder_stack.0.dense.weight", "encoder.node_encoder_stack.2.weight",
              "encoder.masked_lm_pooler.weight", "encoder.lm_head_transform_weight.bias", "encoder.layer_norm.weight",
              "encoder.lm_output_learned_bias", "encoder.embed_out.weight"):
        assert k in sd, k
    ge = model.encoder.graph_encoder
    assert len(ge.fusion_layers) == hp.n_fusion_stacks and len(ge.layers) == hp.n_graph_stacks
    assert len(ge.text_model.encoder.layer) == hp.n_pre_text


def test_qkv_roundtrip_and_upgrade():
    hp, model = build("B")
    fill_hash_weights(model)
    from oracle import hashinit
    ge = model.encoder.graph_encoder
    att = ge.layers[0].layers[0].self_attn
    d = hp.dim
    for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
        ref = hashinit.param(f"layers.0.layers.0.self_attn.{n}.weight", (d, d))
        assert np.array_equal(att.qkv_weight[i * d:(i + 1) * d].detach().numpy(), ref)
        assert np.array_equal(getattr(att, n).weight.detach().numpy(), ref)
    bl = ge.fusion_layers[0].fusion_layers[1].bert_encoder
    ref = hashinit.param("fusion_layers.0.fusion_layers.1.bert_encoder.attention.self.key.bias", (d,))
    assert np.array_equal(bl.attention.self.qkv_bias[d:2 * d].detach().numpy(), ref)
    # legacy fused in_proj checkpoints are split into q/k/v names
    sd = {"x.in_proj_weight": torch.arange(12.).view(6, 2), "x.in_proj_bias": torch.arange(6.)}
    att.upgrade_state_dict_named(sd, "x")
    assert set(sd) == {"x.q_proj.weight", "x.k_proj.weight", "x.v_proj.weight", "x.q_proj.bias", "x.k_proj.bias", "x.v_proj.bias"}
    assert torch.equal(sd["x.k_proj.weight"], torch.arange(12.).view(6, 2)[2:4])


def test_freeze_initial_encoders_matches_reference_rule():
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams("A")
    model = GraphormerModel.build_model(model_args(hp, freeze_initial_encoders=True), task=None)
    ge = model.encoder.graph_encoder
    assert not ge.text_model.embeddings.word_embeddings.weight.requires_grad
    assert not ge.vit_model.layernorm.weight.requires_grad
    assert not ge.text_model.encoder.layer[0].attention.self.qkv_weight.requires_grad
    assert ge.text_model.pooler.dense.weight.requires_grad and ge.node_classifier.weight.requires_grad
    assert ge.fusion_layers[0].fusion_layers[0].bert_encoder.attention.self.qkv_weight.requires_grad


def test_collator_signature_and_parity():
    """``collator(items, spatial_pos_max)`` over per-tree tensors equals the oracle collation."""
    from multimodaldiscussiontransformer_amd.data import collator
    name, trees = cases.structure_specs()[3]
    items = []
    for i, t in enumerate(trees):
        sp, dist, deg = S.preprocess_tree(t["parent"])
        n = len(t["parent"])
        imgs = torch.from_numpy(t["images"]) if t["images"] is not None else torch.zeros(1, 3, 8, 8)
        items.append((i, torch.zeros(n + 1, n + 1), torch.from_numpy(sp).float(), torch.from_numpy(deg),
                      {k: torch.from_numpy(t[k]) for k in ("input_ids", "token_type_ids", "attention_mask")},
                      torch.from_numpy(t["image_index"].astype(np.float32)), imgs, torch.from_numpy(dist).float(),
                      torch.from_numpy(t["y"])))
    for spm in (5, 10):
        out = collator(items, spm)
        ref = S.collate(trees, spm)
        for k in ("attn_bias", "spatial_pos", "in_degree", "x_token_mask", "x", "x_token_type_ids", "x_attention_mask",
                  "x_image_indexes", "y"):
            assert np.array_equal(out[k].numpy(), ref[k]), k
            assert out[k].numpy().dtype == ref[k].dtype, k
        assert out["out_degree"] is out["in_degree"]
        assert tuple(out["x_images"].shape) == ref["x_images"].shape


def test_no_cpu_fallback():
    """The HIP path is the only path: CPU tensors are rejected loudly."""
    from multimodaldiscussiontransformer_amd import ops
    from multimodaldiscussiontransformer_amd._lib import MdtError
    with pytest.raises(MdtError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    hp, model = build("B")
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    pb = pack_batch(cases.tiny_trees("B", hp), 5, device="cpu")
    with pytest.raises(RuntimeError):
        model(pb.batched_data)


def test_f1_metrics_match_oracle():
    from multimodaldiscussiontransformer_amd.criterions import GraphPredictionNodeCrossEntropy as C
    for c in (dict(ncorrect=5, num_positive_correct=2, total_positive=4, num_pred_positive=3, sample_size=9),
              dict(ncorrect=3, num_positive_correct=0, total_positive=0, num_pred_positive=0, sample_size=3),
              dict(ncorrect=0, num_positive_correct=0, total_positive=2, num_pred_positive=1, sample_size=4)):
        want = R.f1_metrics(c)
        got = C.compute_metrics([dict(loss=1.0, **c)])
        for k in ("accuracy", "recall", "precision", "f1"):
            assert abs(got[k] - want[k]) < 1e-12
