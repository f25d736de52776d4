"""hf_weights.py: a HuggingFace BERT / ViT state dict (what the reference's ``from_pretrained`` calls return,
mDT/src/modules/multigraphormer_graph_encoder.py:233-278) lands in the right blocks of the product model — embeddings,
pre-fusion blocks, fusion blocks in execution order, final LayerNorm, poolers with their aliases, classifier — and a
model that is not available locally is an error, never a silent random init."""
import pytest
import torch

from oracle import cases
from tests.util_model import model_args

transformers = pytest.importorskip("transformers")


def _hf_models(hp):
    bc = transformers.BertConfig(hidden_size=hp.dim, num_hidden_layers=hp.text_layers, num_attention_heads=hp.enc_heads,
                                 intermediate_size=hp.enc_ffn, vocab_size=hp.vocab_size, max_position_embeddings=hp.max_pos,
                                 type_vocab_size=hp.type_vocab, num_labels=2)
    vc = transformers.ViTConfig(hidden_size=hp.dim, num_hidden_layers=hp.vit_layers, num_attention_heads=hp.enc_heads,
                                intermediate_size=hp.enc_ffn, image_size=hp.image_size, patch_size=hp.patch)
    torch.manual_seed(3)
    return transformers.BertForSequenceClassification(bc), transformers.ViTModel(vc, add_pooling_layer=True)


@pytest.mark.parametrize("kind", ["A", "B"])
def test_huggingface_weights_land_in_the_right_blocks(kind):
    from multimodaldiscussiontransformer_amd import hf_weights
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams(kind)
    model = GraphormerModel.build_model(model_args(hp), task=None)
    ge = model.encoder.graph_encoder
    bert, vit = _hf_models(hp)
    info = hf_weights.load_pretrained_encoders(ge, bert=bert.state_dict(), vit=vit.state_dict())
    assert info["bert"] and info["vit"] and info["loaded"] > 0
    d = hp.dim
    n_pre = hp.n_pre_text
    hb = bert.bert
    vsd = hf_weights.normalize_vit_keys(vit.state_dict())        # 4.x names whatever transformers is installed
    # embeddings and a pre-fusion block
    assert torch.equal(ge.text_model.embeddings.word_embeddings.weight, hb.embeddings.word_embeddings.weight)
    if n_pre:
        a = ge.text_model.encoder.layer[0].attention.self
        assert torch.equal(a.qkv_weight[d:2 * d], hb.encoder.layer[0].attention.self.key.weight)
    # fusion blocks: HF layer n_pre + i is the i-th fusion layer in execution order
    slots = [(s, j) for s, st in enumerate(ge.fusion_layers) for j in range(len(st.fusion_layers))]
    assert len(slots) == hp.text_layers - n_pre
    for i, (s, j) in enumerate(slots):
        fl = ge.fusion_layers[s].fusion_layers[j]
        src = hb.encoder.layer[n_pre + i]
        assert torch.equal(fl.bert_encoder.attention.self.qkv_weight[:d], src.attention.self.query.weight)
        assert torch.equal(fl.bert_encoder.output.dense.weight, src.output.dense.weight)
        vl = f"encoder.layer.{len(ge.vit_model.encoder.layer) + i}."
        assert torch.equal(fl.vit_encoder.attention.attention.qkv_bias[2 * d:], vsd[vl + "attention.attention.value.bias"])
        assert torch.equal(fl.vit_encoder.layernorm_before.weight, vsd[vl + "layernorm_before.weight"])
        assert torch.equal(fl.vit_encoder.output.dense.weight, vsd[vl + "output.dense.weight"])
    # poolers (one module, two names), classifier, ViT's final LayerNorm and patch projection
    sd = ge.state_dict()
    assert torch.equal(sd["text_pooler.dense.weight"], hb.pooler.dense.weight)
    assert torch.equal(sd["text_model.pooler.dense.weight"], hb.pooler.dense.weight)
    assert torch.equal(sd["vit_pooler.dense.bias"], vsd["pooler.dense.bias"])
    assert torch.equal(ge.node_classifier.weight, bert.classifier.weight)
    assert torch.equal(ge.vit_model.layernorm.weight, vsd["layernorm.weight"])
    assert torch.equal(sd["vit_model.embeddings.patch_embeddings.projection.weight"], vsd["embeddings.patch_embeddings.projection.weight"])
    assert torch.equal(sd["vit_model.embeddings.position_embeddings"], vsd["embeddings.position_embeddings"])


def test_wrong_shape_and_missing_model_are_errors():
    from multimodaldiscussiontransformer_amd import hf_weights
    from multimodaldiscussiontransformer_amd.models import GraphormerModel
    hp = cases.tiny_hparams("A")
    ge = GraphormerModel.build_model(model_args(hp), task=None).encoder.graph_encoder
    bert, _ = _hf_models(hp)
    sd = bert.state_dict()
    sd["bert.embeddings.word_embeddings.weight"] = sd["bert.embeddings.word_embeddings.weight"][:-1]
    with pytest.raises(ValueError, match="does not fit"):
        hf_weights.load_pretrained_encoders(ge, bert=sd)
    sd = bert.state_dict()
    del sd["bert.encoder.layer.0.output.dense.bias"]
    with pytest.raises(ValueError, match="do not cover"):
        hf_weights.load_pretrained_encoders(ge, bert=sd)
    with pytest.raises(FileNotFoundError, match="never downloads"):
        hf_weights.load_pretrained_encoders(ge, bert="some-org/model-that-is-not-in-the-local-cache")
