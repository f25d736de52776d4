"""SURVEY.md §8b under a FairSeq whose registries enforce their base classes.

fairseq is absent from the build image, so the package is imported in a CHILD process whose sys.path carries
tests/fake_fairseq — a stand-in that reproduces the checks of the real registries (``register_model``: must extend
BaseFairseqModel; ``register_task``: must extend FairseqTask; the criterion registry: must extend FairseqCriterion;
dataclasses: must extend FairseqDataclass; duplicate names rejected) and ``import_user_module``'s handling of
``--user-dir``.  Reference: mDT/src/models/multi_modal_discussion_transformer.py:22-23,
mDT/src/tasks/node_prediction.py:22, mDT/src/criterions/hatespeech_loss.py:40-43,
mDT/experiments/hateful_discussions/run_train.sh:29.  Nothing here needs a GPU: construction and registration only."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code: str) -> dict:
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "fake_fairseq"), ROOT, env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_user_dir_src_registers_everything_with_fairseq():
    out = _child(f"""
        import json, sys
        from argparse import Namespace
        import fairseq
        from fairseq import utils, models, tasks, criterions
        assert "standin" in fairseq.__version__
        utils.import_user_module(Namespace(user_dir={os.path.join(ROOT, 'src')!r}))      # fairseq-train --user-dir ../../src
        import src
        from src.models import GraphormerModel                 # criterions/hatespeech_loss.py:18 of the reference
        from src.data import register_dataset, DATASET_REGISTRY   # experiments/.../datasets/dataset.py:1
        import src.tasks.node_prediction, src.modules.multihead_attention, src.data.collator
        import multimodaldiscussiontransformer_amd as pkg
        from multimodaldiscussiontransformer_amd import registry
        assert registry.HAVE_FAIRSEQ
        assert src.models is sys.modules["multimodaldiscussiontransformer_amd.models"]          # one module object, two names
        assert src.tasks.node_prediction is sys.modules["multimodaldiscussiontransformer_amd.tasks.node_prediction"]
        assert issubclass(GraphormerModel, models.FairseqEncoderModel) and issubclass(GraphormerModel, models.BaseFairseqModel)
        from src.models import GraphormerEncoder
        assert issubclass(GraphormerEncoder, models.FairseqEncoder)
        from src.tasks import NodePredictionTask, ContrastiveLearningTask, TaskConfig
        assert issubclass(NodePredictionTask, tasks.FairseqTask) and issubclass(ContrastiveLearningTask, tasks.FairseqTask)
        from src.criterions import GraphPredictionNodeCrossEntropy, GraphContrastiveLoss
        assert issubclass(GraphPredictionNodeCrossEntropy, criterions.FairseqCriterion)
        assert issubclass(GraphContrastiveLoss, criterions.FairseqCriterion)
        from fairseq.dataclass import FairseqDataclass
        assert issubclass(TaskConfig, FairseqDataclass)
        print(json.dumps(dict(models=sorted(models.MODEL_REGISTRY), archs=sorted(models.ARCH_MODEL_REGISTRY),
                              tasks=sorted(tasks.TASK_REGISTRY), criterions=sorted(criterions.CRITERION_REGISTRY),
                              crit_dc=sorted(criterions.CRITERION_DATACLASS_REGISTRY), task_dc=sorted(tasks.TASK_DATACLASS_REGISTRY))))
    """)
    assert out["models"] == ["multi_graphormer"]
    assert out["archs"] == ["multi_graphormer", "multi_graphormer_base"]
    assert out["tasks"] == ["contrastive_learning", "node_prediction"]
    assert out["criterions"] == ["contrastive_loss", "node_cross_entropy"]
    assert out["crit_dc"] == out["criterions"] and out["task_dc"] == out["tasks"]


def test_fairseq_builds_task_model_and_criterion_from_registries():
    """What fairseq_cli.train does with the registries: setup_task → task.build_model (arch function + model class from
    fairseq's tables) → task.build_criterion (constructor arguments by name from the config)."""
    out = _child(f"""
        import json
        from argparse import Namespace
        from fairseq import utils, models, tasks, criterions
        utils.import_user_module(Namespace(user_dir={os.path.join(ROOT, 'src')!r}))
        from src.tasks import NodePredictionConfig
        tiny = dict(dim=128, layers=4, heads=4, intermediate=128)
        cfg = NodePredictionConfig(dataset_name="none", spatial_pos_max=5, max_nodes=64)
        cfg.task = "node_prediction"
        task = tasks.setup_task(cfg)
        assert task.max_nodes() == 64 and task.source_dictionary is None and task.target_dictionary is None
        args = Namespace(arch="multi_graphormer_base", num_bottleneck_tokens=2, num_fusion_layers=1, encoder_embed_dim=128,
                         encoder_ffn_embed_dim=128, encoder_attention_heads=4, dropout=0.1, attention_dropout=0.1, act_dropout=0.1,
                         bert_config=dict(tiny, vocab=512, max_pos=32, type_vocab=2), vit_config=dict(tiny, image_size=32, patch=16))
        model = task.build_model(args)
        assert type(model).__name__ == "GraphormerModel" and isinstance(model, models.BaseFairseqModel)
        assert isinstance(model.encoder, models.FairseqEncoder) and args.max_nodes == 64
        keys = list(model.state_dict())
        assert any(k.startswith("node_encoder_stack.2.") for k in keys)              # tasks/node_prediction.py:47-53
        ccfg = Namespace(criterion="node_cross_entropy", positive_weight=1.5, negative_weight=1.0)
        crit = task.build_criterion(ccfg)
        assert crit.positive_weight == 1.5 and crit.task is task and crit.logging_outputs_can_be_summed()
        c2 = criterions.build_criterion(Namespace(criterion="contrastive_loss", soft_negative_weight=0.0, multiplication_scale=20.0,
                                                  adaptive_soft_negative_weight=True), task)
        assert c2.multiplication_scale == 20.0
        try:
            criterions.build_criterion(Namespace(criterion="contrastive_loss", soft_negative_weight=0.5, multiplication_scale=20.0,
                                                 adaptive_soft_negative_weight=True), task)
            raise SystemExit("mutually exclusive flags must raise")
        except ValueError:
            pass
        print(json.dumps(dict(nkeys=len(keys))))
    """)
    assert out["nkeys"] > 100


def test_registries_reject_what_fairseq_rejects():
    out = _child("""
        import json
        import torch.nn as nn
        from multimodaldiscussiontransformer_amd import registry as R
        from multimodaldiscussiontransformer_amd import models, tasks, criterions    # first registration
        errs = []
        for what, fn in (("plain nn.Module as model", lambda: R.register_model("x_model")(type("M", (nn.Module,), {}))),
                         ("plain object as task", lambda: R.register_task("x_task")(type("T", (object,), {}))),
                         ("plain nn.Module as criterion", lambda: R.register_criterion("x_crit")(type("C", (nn.Module,), {}))),
                         ("arch of unknown model", lambda: R.register_model_architecture("nope", "x_arch")(lambda a: a)),
                         ("duplicate model name", lambda: R.register_model("multi_graphormer")(models.GraphormerModel))):
            try:
                fn()
                errs.append(what)
            except ValueError:
                pass
        print(json.dumps(dict(not_rejected=errs)))
    """)
    assert out["not_rejected"] == []


def test_same_names_resolve_without_fairseq():
    import multimodaldiscussiontransformer_amd.criterions  # noqa: F401
    import multimodaldiscussiontransformer_amd.models  # noqa: F401
    import multimodaldiscussiontransformer_amd.tasks  # noqa: F401
    from multimodaldiscussiontransformer_amd import registry as R
    assert not R.HAVE_FAIRSEQ
    assert sorted(R.MODEL_REGISTRY) == ["multi_graphormer"]
    assert sorted(R.ARCH_CONFIG_REGISTRY) == ["multi_graphormer", "multi_graphormer_base"]
    assert sorted(R.TASK_REGISTRY) == ["contrastive_learning", "node_prediction"]
    assert sorted(R.CRITERION_REGISTRY) == ["contrastive_loss", "node_cross_entropy"]
    cls, dc = R.CRITERION_REGISTRY["node_cross_entropy"]
    assert issubclass(cls, R.FairseqCriterion) and issubclass(dc, R.FairseqDataclass)
