"""Drop-in for mDT/src/models/multi_modal_discussion_transformer.py: model ``multi_graphormer``
with architectures ``multi_graphormer`` / ``multi_graphormer_base``, the reference's CLI flags
(underscore flags included) and the classifier head — pooler → dropout → classifier applied to
both the text [CLS] and bottleneck token 0 with the *same* weights, averaged (:265-274).

Encoder, head and their backward run as ONE tape (one autograd node); with
``prepare_main_grads`` parameter gradients accumulate in persistent fp32 buffers laid out in a
single flat arena, which is what the RCCL data-parallel hook all-reduces.
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

from .. import engine as E
from ..data.packer import packed_from_batched_data
from ..modules import MultiGraphormerGraphEncoder, init_graphormer_params
from ..registry import FairseqEncoder, FairseqEncoderModel, register_model, register_model_architecture

logger = logging.getLogger(__name__)


def _slot(n: int) -> int:
    """arena slot size: 256-byte aligned so every main_grad view is vector-addressable"""
    return (n + 63) // 64 * 64


def safe_hasattr(obj, k):
    return getattr(obj, k, None) is not None


@register_model("multi_graphormer")
class GraphormerModel(FairseqEncoderModel):
    """mDT/src/models/multi_modal_discussion_transformer.py:22-178 (extends FairseqEncoderModel there as here)."""

    def __init__(self, args, encoder):
        super().__init__(encoder)
        self.args = args
        if getattr(args, "apply_graphormer_init", False):
            self.apply(init_graphormer_params)
        self.encoder_embed_dim = args.encoder_embed_dim

    @staticmethod
    def add_args(parser):
        """Model-specific arguments, names identical to the reference (:33-158)."""
        parser.add_argument("--dropout", type=float, metavar="D", help="dropout probability")
        parser.add_argument("--attention-dropout", type=float, metavar="D", help="dropout probability for attention weights")
        parser.add_argument("--act-dropout", type=float, metavar="D", help="dropout probability after activation in FFN")
        parser.add_argument("--encoder-ffn-embed-dim", type=int, metavar="N", help="encoder embedding dimension for FFN")
        parser.add_argument("--encoder-layers", type=int, metavar="N", help="num encoder layers")
        parser.add_argument("--encoder-attention-heads", type=int, metavar="N", help="num encoder attention heads")
        parser.add_argument("--num_fusion_layers", type=int, metavar="N", help="num fusion layers")
        parser.add_argument("--num_graph_stack", type=int, metavar="N", help="num graph layers per fusion layer")
        parser.add_argument("--num_fusion_stack", type=int, metavar="N", help="num fusion layers per graph layer")
        parser.add_argument("--num_bottleneck_tokens", type=int, metavar="N", help="num bottleneck tokens")
        parser.add_argument("--encoder-embed-dim", type=int, metavar="N", help="encoder embedding dimension")
        parser.add_argument("--split", type=int, metavar="N", help="dataset split to use (not used in code)")
        parser.add_argument("--share-encoder-input-output-embed", action="store_true",
                            help="share encoder input and output embeddings")
        parser.add_argument("--encoder-learned-pos", action="store_true",
                            help="use learned positional embeddings in the encoder")
        parser.add_argument("--no-token-positional-embeddings", action="store_true",
                            help="if set, disables positional embeddings (outside self attention)")
        parser.add_argument("--max-positions", type=int, help="number of positional embeddings to learn")
        parser.add_argument("--apply-graphormer-init", action="store_true",
                            help="use custom param initialization for Graphormer")
        parser.add_argument("--activation-fn", choices=["relu", "gelu"], help="activation function to use")
        parser.add_argument("--encoder-normalize-before", action="store_true",
                            help="apply layernorm before each encoder block")
        parser.add_argument("--pre-layernorm", action="store_true",
                            help="apply layernorm before self-attention and ffn. Without this, post layernorm will used")
        parser.add_argument("--freeze_initial_encoders", action="store_true", help="freezes the initial layers ")
        # not in the reference, which hard-codes the hub names and downloads (multigraphormer_graph_encoder.py:236-245):
        parser.add_argument("--pretrained-bert", default=None, help="HuggingFace BERT weights for the text stack: local directory / "
                            "file or a name in the LOCAL HF cache (default bert-base-uncased); never downloaded")
        parser.add_argument("--pretrained-vit", default=None, help="same for the image stack (default google/vit-base-patch16-224)")
        parser.add_argument("--random-init-encoders", action="store_true", default=False,
                            help="build the BERT / ViT stacks from random weights instead")

    def max_nodes(self):
        return self.encoder.max_nodes

    @classmethod
    def build_model(cls, args, task):
        base_architecture(args)
        if not safe_hasattr(args, "max_nodes"):
            args.max_nodes = getattr(args, "tokens_per_sample", 10000)
        logger.info(args)
        encoder = GraphormerEncoder(args)
        model = cls(args, encoder)
        # A parser that went through add_args (FairSeq's, or train.py's) carries --random-init-encoders: such a run gets the
        # reference's pretrained encoders unless it opts out or runs custom BERT / ViT shapes; programmatic construction
        # (tests, bench.py: a bare namespace) never touches the disk.
        custom = getattr(args, "bert_config", None) is not None or getattr(args, "vit_config", None) is not None
        if hasattr(args, "random_init_encoders") and not args.random_init_encoders and not custom and \
                not getattr(args, "_skip_pretrained", False):
            from .. import hf_weights
            info = hf_weights.load_pretrained_encoders(encoder.graph_encoder,
                                                       bert=getattr(args, "pretrained_bert", None) or hf_weights.BERT_NAME,
                                                       vit=getattr(args, "pretrained_vit", None) or hf_weights.VIT_NAME)
            logger.info("pretrained encoders: %s", info)
        return model

    def forward(self, batched_data, **kwargs):
        return self.encoder(batched_data, **kwargs)

    # ---- fp32 gradient arena ----------------------------------------------------------------
    def prepare_main_grads(self):
        """Give every trainable parameter an fp32 ``main_grad`` view into one flat buffer (reverse
        registration order ≈ the order gradients become final in backward, so data-parallel buckets
        are contiguous slices).  Returns the flat buffer."""
        params = [p for p in self.encoder.graph_encoder.live_parameters() if p.requires_grad]
        extra = [p for p in self.parameters() if p.requires_grad and all(p is not q for q in params)]
        params = params + extra
        total = sum(_slot(p.numel()) for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        off = 0
        for p in reversed(params):
            p.main_grad = flat[off:off + p.numel()].view(p.shape)
            off += _slot(p.numel())
        self.main_grad_flat = flat
        self.encoder.graph_encoder.use_main_grad = True
        return flat

    def zero_main_grads(self):
        self.main_grad_flat.zero_()

    def enable_fp8(self, on: bool = True, sites=None):
        """BASELINE.json configs[4]: per-tensor-scaled fp8 operands (e4m3 activations / weights, e5m2 gradients, delayed
        scaling) for the encoder blocks' big GEMMs.  ``sites``: a preset of fp8.PRESETS ("all" — the default —, "fast4",
        "grads") or a comma list of site names (fp8.py says which GEMMs and what each choice costs in accuracy)."""
        from .. import fp8
        fp8.ACTIVE = fp8.Fp8State(next(self.parameters()).device, sites=sites) if on else None
        return fp8.ACTIVE


class GraphormerEncoder(FairseqEncoder):
    def __init__(self, args):
        super().__init__(dictionary=None)
        self.max_nodes = args.max_nodes
        extra = {}
        if getattr(args, "bert_config", None) is not None:
            extra["bert_config"] = args.bert_config
        if getattr(args, "vit_config", None) is not None:
            extra["vit_config"] = args.vit_config
        self.graph_encoder = MultiGraphormerGraphEncoder(
            num_atoms=args.num_atoms, num_in_degree=args.num_in_degree, num_out_degree=args.num_out_degree,
            num_edges=args.num_edges, num_spatial=args.num_spatial, num_edge_dis=args.num_edge_dis,
            edge_type=args.edge_type, multi_hop_max_dist=args.multi_hop_max_dist,
            num_bottle_neck=args.num_bottleneck_tokens, num_fusion_layers=args.num_fusion_layers,
            num_fusion_stack=args.num_fusion_stack, num_graph_stack=args.num_graph_stack,
            num_encoder_layers=args.encoder_layers, embedding_dim=args.encoder_embed_dim,
            ffn_embedding_dim=args.encoder_ffn_embed_dim, num_attention_heads=args.encoder_attention_heads,
            dropout=args.dropout, attention_dropout=args.attention_dropout, activation_dropout=args.act_dropout,
            encoder_normalize_before=args.encoder_normalize_before, pre_layernorm=args.pre_layernorm,
            apply_graphormer_init=args.apply_graphormer_init, activation_fn=args.activation_fn,
            freeze_initial_encoders=args.freeze_initial_encoders, **extra)
        ge = self.graph_encoder
        # the same module objects under a second name, exactly as the reference registers them (:215-221)
        self.node_encoder_stack = nn.ModuleList([ge.text_pooler, ge.text_dropout, ge.node_classifier])
        self.share_input_output_embed = args.share_encoder_input_output_embed
        self.embed_out = None
        self.lm_output_learned_bias = None
        self.load_softmax = not getattr(args, "remove_head", False)
        # dead parameters of the reference head (:231-247), kept for checkpoint compatibility
        self.masked_lm_pooler = nn.Linear(args.encoder_embed_dim, args.encoder_embed_dim)
        self.lm_head_transform_weight = nn.Linear(args.encoder_embed_dim, args.encoder_embed_dim)
        self.layer_norm = nn.LayerNorm(args.encoder_embed_dim)
        if self.load_softmax:
            self.lm_output_learned_bias = nn.Parameter(torch.zeros(1))
            if not self.share_input_output_embed:
                self.embed_out = nn.Linear(args.encoder_embed_dim, args.num_classes, bias=False)
            else:
                raise NotImplementedError

    def forward(self, batched_data, **unused):
        """→ (logits [M, 2], global_embedding [B, D])."""
        ge = self.graph_encoder
        pb = packed_from_batched_data(batched_data)
        p_head = ge.text_dropout.p if self.training else 0.0
        nb = ge.num_bottle_neck

        def run(tape):
            text, glob, rows = ge._fwd(tape, pb, prune_last=ge.prune_last_layer)
            logits = E.classifier_head(tape, text, pb.M, rows["cls_rows"], rows["bn0_rows"], ge.text_pooler.dense.weight,
                                       ge.text_pooler.dense.bias, ge.node_classifier.weight, ge.node_classifier.bias,
                                       p_drop=p_head)
            return logits, glob

        logits, glob = E.run_tape(run, [], ge.live_parameters(), use_main_grad=ge.use_main_grad, hook=ge.grad_ready_hook)
        return logits, glob

    def upgrade_state_dict_named(self, state_dict, name):
        if not self.load_softmax:
            for k in list(state_dict.keys()):
                if "embed_out.weight" in k or "lm_output_learned_bias" in k:
                    del state_dict[k]
        return state_dict


@register_model_architecture("multi_graphormer", "multi_graphormer")
def base_architecture(args):
    args.dropout = getattr(args, "dropout", 0.1)
    args.attention_dropout = getattr(args, "attention_dropout", 0.1)
    args.act_dropout = getattr(args, "act_dropout", 0.0)
    args.encoder_ffn_embed_dim = getattr(args, "encoder_ffn_embed_dim", 4096)
    args.encoder_layers = getattr(args, "encoder_layers", 6)
    args.encoder_attention_heads = getattr(args, "encoder_attention_heads", 8)
    args.split = getattr(args, "split", 0)
    args.encoder_embed_dim = getattr(args, "encoder_embed_dim", 1024)
    args.share_encoder_input_output_embed = getattr(args, "share_encoder_input_output_embed", False)
    args.no_token_positional_embeddings = getattr(args, "no_token_positional_embeddings", False)
    args.num_bottleneck_tokens = getattr(args, "num_bottleneck_tokens", 4)
    args.num_fusion_layers = getattr(args, "num_fusion_layers", 4)
    args.num_graph_stack = getattr(args, "num_graph_stack", 1)
    args.num_fusion_stack = getattr(args, "num_fusion_stack", 1)
    args.apply_graphormer_init = getattr(args, "apply_graphormer_init", False)
    args.activation_fn = getattr(args, "activation_fn", "gelu")
    args.encoder_normalize_before = getattr(args, "encoder_normalize_before", True)
    # task-level defaults (mDT/src/tasks/task.py:29-113) for callers that build the model directly
    for k, v in dict(num_atoms=512 * 9, num_in_degree=512, num_out_degree=512, num_edges=512 * 3, num_spatial=512,
                     num_edge_dis=128, edge_type="multi_hop", multi_hop_max_dist=5, num_classes=1, pre_layernorm=False,
                     freeze_initial_encoders=False).items():
        if not hasattr(args, k):
            setattr(args, k, v)


@register_model_architecture("multi_graphormer", "multi_graphormer_base")
def graphormer_base_architecture(args):
    args.encoder_embed_dim = getattr(args, "encoder_embed_dim", 80)
    args.split = getattr(args, "split", 0)
    args.encoder_layers = getattr(args, "encoder_layers", 12)
    args.encoder_attention_heads = getattr(args, "encoder_attention_heads", 8)
    args.encoder_ffn_embed_dim = getattr(args, "encoder_ffn_embed_dim", 80)
    args.activation_fn = getattr(args, "activation_fn", "gelu")
    args.freeze_initial_encoders = getattr(args, "freeze_initial_encoders", False)
    args.encoder_normalize_before = getattr(args, "encoder_normalize_before", True)
    args.apply_graphormer_init = getattr(args, "apply_graphormer_init", False)
    args.share_encoder_input_output_embed = getattr(args, "share_encoder_input_output_embed", False)
    args.no_token_positional_embeddings = getattr(args, "no_token_positional_embeddings", False)
    args.pre_layernorm = getattr(args, "pre_layernorm", False)
    base_architecture(args)
