"""world_size-2 gloo test of the data-parallel gradient exchange (the same GradientBucketer the
GPU path drives with RCCL): bucketed in-place all-reduce of the flat fp32 arena in completion
order, completion-order re-layout after the first step, parameters that never report, and the
FairSeq-convention scaling by the global sample size."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multimodaldiscussiontransformer_amd.ddp import GradientBucketer
        torch.manual_seed(0)
        shapes = [(64, 8), (8,), (128, 16), (16,), (32, 32), (5,), (300,)]
        params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        params[5].requires_grad = False                       # frozen: never enters the arena
        train = [p for p in params if p.requires_grad]
        flat = torch.zeros(sum(p.numel() for p in train))
        b = GradientBucketer(params, flat, bucket_bytes=4 * 600)      # small buckets → several all-reduces
        fire_order = [[train[4], train[2]], [train[3]], [train[0], train[1]]]   # train[5] (the [300] vector) never reports
        if rank == 1:
            # this rank's batches have no image comments: one group never reports and the order differs — the
            # arena layout and the sequence of collectives must still be the ones rank 0 derives
            fire_order = [[train[3]], [train[4], train[2]]]
        results = []
        for step in range(3):
            flat.zero_()
            g = torch.Generator().manual_seed(100 * step + rank)
            local = {}
            for p in train:
                v = torch.randn(p.shape, generator=g)
                p.main_grad.copy_(v)
                local[id(p)] = v
            for grp in fire_order:
                b.on_params_ready(grp)
            scal = torch.tensor([1.5 + rank, float(2 + rank), 1.0, 0.0, 1.0, 1.0])
            b.finish(scal)
            if not b.layout_final:
                b.finalize_layout()
            # expected: sum over ranks / global sample size
            tot = float(sum(2 + r for r in range(world)))
            for p in train:
                exp = torch.zeros(p.shape)
                for r in range(world):
                    gr = torch.Generator().manual_seed(100 * step + r)
                    for q2 in train:
                        v = torch.randn(q2.shape, generator=gr)
                        if q2 is p:
                            exp += v
                torch.testing.assert_close(p.main_grad, exp / tot, atol=1e-6, rtol=1e-6)
            assert abs(float(scal[1]) - tot) < 1e-6 and abs(float(scal[0]) - sum(1.5 + r for r in range(world))) < 1e-6
            results.append(True)
        # after re-layout the arena starts with rank 0's first-finished group and ends with the silent parameter
        first = b.slots[0][0]
        assert first == id(train[4]) and b.slots[-1][0] == id(train[5])
        assert len(b.bucket_ends) >= 2 and b.bucket_ends[-1] == flat.numel()
        q.put((rank, "ok", len(results)))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gradient_bucketer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in out:
        assert status == "ok", f"rank {rank}: {info}"


def test_relayout_preserves_gradients_single_process():
    from multimodaldiscussiontransformer_amd.ddp import GradientBucketer
    params = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2))]
    flat = torch.zeros(12 + 5 + 4)
    b = GradientBucketer(params, flat)
    for i, p in enumerate(params):
        p.main_grad.fill_(float(i + 1))
    b.on_params_ready([params[2]])
    b.on_params_ready([params[0]])
    b.finish(None)
    b.finalize_layout()
    assert [s[0] for s in b.slots] == [id(params[2]), id(params[0]), id(params[1])]
    for i, p in enumerate(params):
        assert torch.all(p.main_grad == float(i + 1))
        assert p.main_grad.data_ptr() >= flat.data_ptr()


def test_folded_gradient_scale_equals_the_arena_divide():
    """finish(fold_scale=True) leaves the arena holding the sum and hands back 1 / max(sample size, 1) for the optimiser:
    sum * scale is what the unfolded path writes into the arena (1 ulp: a multiply by the reciprocal instead of a divide)."""
    from multimodaldiscussiontransformer_amd.ddp import GradientBucketer
    out = {}
    for fold in (False, True):
        params = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(5))]
        flat = torch.zeros(26)
        b = GradientBucketer(params, flat)
        torch.manual_seed(3)
        flat.copy_(torch.randn(26))
        scal = torch.tensor([1.5, 6.0, 0, 0, 0, 0])
        scale = b.finish(scal, fold_scale=fold)
        out[fold] = (flat.clone(), scale)
    assert out[False][1] is None
    assert float(out[True][1]) == pytest.approx(1.0 / 6.0)
    torch.manual_seed(3)
    assert torch.equal(out[True][0], torch.randn(26))                      # untouched
    assert torch.allclose(out[True][0] * out[True][1], out[False][0], rtol=3e-7, atol=0)
    b = GradientBucketer([torch.nn.Parameter(torch.zeros(2))], torch.zeros(2))
    assert float(b.finish(torch.tensor([0.0, 0.0]), fold_scale=True)) == 1.0      # an empty batch divides by 1, as before


def test_balance_trees_by_token_cost():
    """greedy assignment by N_i (L + nb) + I_i (P + nb).  Skewed image fractions: dealing trees round-robin
    leaves one rank with most of the image work; the balanced deal is within a few per cent."""
    from multimodaldiscussiontransformer_amd.ddp import balance_trees
    import numpy as np
    rng = np.random.default_rng(3)
    n = rng.integers(32, 65, size=64).tolist()
    img = [int(k * f) for k, f in zip(n, np.where(np.arange(64) % 8 == 0, 0.9, 0.05))]      # every 8th tree is image-heavy
    cost = [a * 104 + b * 201 for a, b in zip(n, img)]
    world = 8
    share = balance_trees(n, img, world)
    assert sorted(k for sh in share for k in sh) == list(range(64)) and all(sh == sorted(sh) for sh in share)
    load = [sum(cost[k] for k in sh) for sh in share]
    naive = [sum(cost[k] for k in range(r, 64, world)) for r in range(world)]
    assert max(load) / (sum(load) / world) < 1.03, load
    assert max(naive) / (sum(naive) / world) > 1.5, naive          # what seeding trees by rank would have done here
    assert balance_trees(n, img, world) == share                      # deterministic: every rank computes the same deal
    assert balance_trees([5], [0], 2) == [[0], []]
    eq = balance_trees([10] * 6, [0] * 6, 3)
    assert [len(s) for s in eq] == [2, 2, 2]


def _worker_wire_and_broadcast(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multimodaldiscussiontransformer_amd.ddp import DataParallel, GradientBucketer
        # (a) bf16 on the wire: each bucket travels as a bf16 copy and comes back into the fp32 arena
        params = [torch.nn.Parameter(torch.zeros(s)) for s in ((40, 8), (8,), (64, 4))]
        flat = torch.zeros(sum(p.numel() for p in params))
        b = GradientBucketer(params, flat, bucket_bytes=4 * 300, wire_dtype=torch.bfloat16)
        for step in range(2):
            flat.zero_()
            for i, p in enumerate(params):
                p.main_grad.fill_(0.5 * (i + 1) * (rank + 1))            # exactly representable in bf16
            for p in params:
                b.on_params_ready([p])
            b.finish(torch.tensor([0.0, 1.0, 0, 0, 0, 0]))
            if not b.layout_final:
                b.finalize_layout()
            for i, p in enumerate(params):
                want = 0.5 * (i + 1) * sum(r + 1 for r in range(world)) / world
                assert torch.all(p.main_grad == want), (i, float(p.main_grad.flatten()[0]), want)
            assert flat.dtype == torch.float32
        # (b) the REAL model through DataParallel: one flat broadcast per dtype / chunk makes every rank equal to rank 0
        from types import SimpleNamespace
        from multimodaldiscussiontransformer_amd.models import GraphormerModel
        tiny = dict(dim=128, layers=4, heads=4, intermediate=128)
        args = SimpleNamespace(num_bottleneck_tokens=2, num_fusion_layers=1, encoder_embed_dim=128, encoder_ffn_embed_dim=128,
                               encoder_attention_heads=4, dropout=0.0, attention_dropout=0.0, act_dropout=0.0,
                               bert_config=dict(tiny, vocab=512, max_pos=32, type_vocab=2), vit_config=dict(tiny, image_size=32, patch=16))
        torch.manual_seed(1000 + rank)                                  # different weights on every rank
        model = GraphormerModel.build_model(args, task=None)
        dp = DataParallel(model, bucket_mb=1)
        before = model.encoder.graph_encoder.bottle_neck.weight.detach().clone()
        dp.broadcast_parameters()
        n_tensors = len({id(t) for t in list(model.parameters()) + list(model.buffers())})
        assert 1 <= dp.broadcast_calls <= 3 and n_tensors > 100, (dp.broadcast_calls, n_tensors)
        digest = torch.stack([p.detach().double().sum() for p in model.parameters()])
        both = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(both, digest)
        assert torch.equal(both[0], both[1])
        if rank == 1:
            assert not torch.equal(before, model.encoder.graph_encoder.bottle_neck.weight.detach())
        d = dp.diagnostics()
        assert d["arena_mb"] > 1 and d["wire_dtype"] == "float32" and d["broadcast_calls"] == dp.broadcast_calls
        q.put((rank, "ok", 0))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_bf16_wire_and_flat_broadcast_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_wire_and_broadcast, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in out:
        assert status == "ok", f"rank {rank}: {info}"


def test_bench_workload_per_rank_is_the_same_at_every_world_size():
    """the driver computes scaling efficiency from bench.py's N = 1, 2, 4, 8 lines, so the default
    workload must give every rank the SAME per-GPU batch at every N (BASELINE.json configs[2]: 256 trees global = 32 per
    GPU, 64 comments each).  --variable-trees is the opt-in balancing demonstration."""
    import bench
    kw = dict(trees_per_gpu=4, nodes=16, image_frac=0.25, image_size=32, patch=16, shape="bushy")
    one = bench.trees_for_rank(3, 0, 1, **kw)
    n1 = sum(len(t["parent"]) for t in one)
    i1 = sum(int(t["image_index"].sum()) for t in one)
    tok1 = sum(int(t["attention_mask"].sum()) for t in one)
    assert n1 == 4 * 16 and i1 == 4 * 4
    for world in (2, 4, 8):
        shares = [bench.trees_for_rank(3, r, world, **kw) for r in range(world)]
        assert [len(s) for s in shares] == [4] * world
        assert [sum(len(t["parent"]) for t in s) for s in shares] == [n1] * world
        assert [sum(int(t["image_index"].sum()) for t in s) for s in shares] == [i1] * world
        # ragged text: valid-token counts differ per tree (U{8..100}) but the expectation per rank is the same
        tok = [sum(int(t["attention_mask"].sum()) for t in s) for s in shares]
        assert max(tok) / min(tok) < 1.35 and abs(sum(tok) / world / tok1 - 1.0) < 0.35
        # disjoint cover of the global batch
        ids = sorted(id(t["input_ids"]) for s in shares for t in s)
        assert len(set(ids)) == 4 * world
    var = [bench.trees_for_rank(3, r, 2, variable=True, **kw) for r in range(2)]
    assert sum(len(t["parent"]) for s in var for t in s) < 2 * n1          # opt-in: smaller trees, NOT the N = 1 workload


def _worker_verify(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from multimodaldiscussiontransformer_amd.ddp import DataParallel, GradientBucketer
        # a stand-in "model": DataParallel only needs prepare_main_grads / parameters / the encoder hook slot
        torch.manual_seed(5)
        params = [torch.nn.Parameter(torch.randn(s)) for s in ((50, 8), (8,), (64, 4), (300,), (7, 7))]

        class M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.ps = torch.nn.ParameterList(params)
                self.encoder = SimpleNamespace(graph_encoder=SimpleNamespace(grad_ready_hook=None))

            def prepare_main_grads(self):
                self.main_grad_flat = torch.zeros(sum(p.numel() for p in params))
                off = 0
                for p in params:
                    p.main_grad = self.main_grad_flat[off:off + p.numel()].view(p.shape)
                    off += p.numel()
                return self.main_grad_flat

        model = M()
        dp = DataParallel(model, bucket_mb=1)
        dp.bucketer.bucket_elems = 200                      # several buckets
        hook = model.encoder.graph_encoder.grad_ready_hook

        def run_step():
            dp.zero_grad()
            g = torch.Generator().manual_seed(77 + rank)
            for p in reversed(params):                      # "backward": last parameter first
                p.main_grad.add_(torch.randn(p.shape, generator=g))
                hook([p])
            dp.finish_backward(torch.tensor([0.0, 2.0, 0, 0, 0, 0]))

        run_step()                                          # first step: whole arena, then the completion-ordered layout
        assert dp.bucketer.layout_final and len(dp.bucketer.bucket_ends) >= 3
        res = dp.verify_exchange(run_step)
        assert res["ok"] and res["replicas_equal"] and res["overlapped_launches"] == len(dp.bucketer.bucket_ends), res
        assert res["overlapped_vs_deferred_rel_l2"] == 0.0 and res["bucketed_vs_flat_rel_l2"] < 1e-6, res
        assert not dp.bucketer.defer
        # a rank that issues its buckets in another order sums mismatched slices: the replicas disagree and the check says so
        dp.zero_grad()
        for p in params:
            p.main_grad.fill_(1.0 + rank)
        dp.bucketer.flat[:4].fill_(5.0 if rank == 0 else 1.0)          # emulate a mismatched reduction result on one rank
        chk = dp.replicas_checksum()
        assert not chk["replicas_equal"] and chk["checksum_max_abs_diff"] > 0
        d = dp.diagnostics()
        assert d["world_seen_by_backend"] == world and d["backend_seen"] == "gloo" and d["launches_last_step"] >= 1
        q.put((rank, "ok", 0))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_exchange_self_check_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_verify, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, info in out:
        assert status == "ok", f"rank {rank}: {info}"


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_n_as_typed_starts_n_ranks(n):
    """`python3 bench.py --gpus N` — the form the driver's N = 1 line uses, typed with N > 1 and no RANK in the environment —
    must start its own ranks (a fresh torch.distributed.run child, bench._self_launch) instead of asking to be launched.
    MDT_BENCH_LAUNCH_ONLY=1 stops every rank after the process group has counted itself (no GPU here)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MDT_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["distributed"]["world_seen_by_backend"] == n and out["steps"] == 3
    if n != 2:
        return
    # a failing rank's exit code comes back through the launcher
    env["MDT_BENCH_LAUNCH_ONLY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus=2", "--config", "nonsense"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
