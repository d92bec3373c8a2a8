"""World-size-2 gloo tests (CPU) of the sharded-scoring plumbing: contiguous shard bounds, the all-gather of per-rank
logits (equal and ragged shards), rank-order concatenation.  The per-rank scorer here is a stand-in row function -
the collective and shard arithmetic are what is under test; the GPU path itself is covered by the -m gpu suite."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from leclip_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _score(x):  # deterministic per-row "logits": depends only on the row, like the real scorer
    return torch.stack([x.flatten(1).sum(1) * (c + 1) for c in range(5)], dim=1)


def _worker(rank, world, port, n_global, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(5)
    images = torch.randn(n_global, 3, 4, 4, generator=g)
    sc = parallel.ShardedScorer(_score, n_out=5)
    lo, hi = parallel.shard_bounds(n_global, rank, world)
    gathered = sc.score_local(images[lo:hi].contiguous(), n_global)
    gathered2 = sc.score_global(images)
    ref = _score(images)
    ok = torch.allclose(gathered, ref) and torch.equal(gathered, gathered2) and gathered.shape == (n_global, 5)
    out_q.put((rank, bool(ok), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [8, 7, 2, 1])
def test_sharded_scoring_world2(n_global):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    (lo0, hi0), (lo1, hi1) = res[0][2], res[1][2]
    assert lo0 == 0 and hi0 == lo1 and hi1 == n_global and (hi0 - lo0) - (hi1 - lo1) in (0, 1)


def test_shard_bounds_cover_exactly():
    for n in range(0, 40):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    x = torch.randn(6, 3, 4, 4)
    sc = parallel.ShardedScorer(_score)
    assert sc.world == 1 and torch.equal(sc.score_local(x), _score(x)) and torch.equal(parallel.all_gather_rows(_score(x)), _score(x))


# ------------------------------------------------------------------------------ data-parallel prompt tuning (N > 1 ranks)
def _trainer_cfg():
    from leclip_amd.config import get_cfg_default
    cfg = get_cfg_default()
    cfg.merge_from_list(["MODEL.BACKBONE.NAME", "tiny", "MODEL.BACKBONE.PATH", "synthetic:1:cond", "INPUT.SIZE", "(32, 32)",
                         "TRAINER.Caption.PREC", "fp32", "OPTIM.MAX_EPOCH", "2", "OPTIM.WARMUP_EPOCH", "0"])
    return cfg


def _tune_worker(rank, world, port, outdir, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    from leclip_amd.registry import build_trainer
    from leclip_amd.train_caption import SyntheticCaptionLoader
    torch.manual_seed(100 + rank)            # ranks start from DIFFERENT random prompts (SEED = -1 in the reference's default cfg)
    tr = build_trainer(_trainer_cfg())
    model = tr.model_default
    ctx_after_build = model.prompt_learner.ctx.detach().clone()
    # the gradient exchange: every rank contributes rank-dependent gradients, all end with the mean, in one flat collective
    params = [p for p in model.prompt_learner.parameters() if p.requires_grad]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    tr._allreduce_grads(params)
    want = [(1 + 2) / 2.0 * (i + 1) for i in range(len(params))]
    grads_ok = all(bool(torch.allclose(p.grad, torch.full_like(p, w))) for p, w in zip(params, want))
    # rank-sharded caption order: same permutation on every rank, disjoint contiguous slices
    caps = torch.arange(23 * 77).view(23, 77)
    loader = SyntheticCaptionLoader(caps, torch.zeros(23, 80), batch_size=5, seed=3, rank=rank, world=world)
    loader.set_epoch(4)
    idx = loader.indices()
    # checkpoints: rank 0 only
    tr.save_model(0, outdir)
    dist.barrier()
    wrote = os.path.exists(os.path.join(outdir, "default", "model.pth.tar-1"))
    out_q.put((rank, ctx_after_build.numpy(), grads_ok, idx.numpy(), wrote, len(params)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_tuning_plumbing_world2(tmp_path):
    """WORLD_SIZE = 2 (gloo): prompts broadcast from rank 0 at construction (ranks draw different random contexts otherwise),
    ONE flat all-reduce gives every rank the mean gradient, the caption sampler hands each rank a disjoint slice of the same
    permutation, and only rank 0 writes checkpoints."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tune_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, ctx0, g0, i0, w0, n0), (_, ctx1, g1, i1, w1, n1) = res
    assert np.array_equal(ctx0, ctx1) and g0 and g1 and n0 == n1 == 6
    assert w0 and w1 and sorted(os.listdir(tmp_path / "default")) == ["checkpoint", "model.pth.tar-1"]
    both = np.concatenate([i0, i1])
    assert len(i0) == len(i1) == 12 and set(both.tolist()) == set(range(23)) and len(set(i0.tolist()) & set(i1.tolist())) <= 1


def test_resume_restores_optimizer_and_scheduler(tmp_path):
    """save_model / resume_model_if_exist (dassl/engine/trainer.py:119-170, torchtools.py:27-82, 126-165): the prompt learner, SGD
    momentum buffers and the cosine schedule position survive a restart; the epoch to continue from is returned."""
    from leclip_amd.registry import build_trainer
    torch.manual_seed(0)
    tr = build_trainer(_trainer_cfg())
    params = [p for p in tr.model_default.prompt_learner.parameters() if p.requires_grad]
    for step in range(3):
        for p in params:
            p.grad = torch.ones_like(p) * (step + 1)
        tr.optim.step()
    tr.update_lr()
    tr.save_model(0, str(tmp_path))
    lr_saved = tr.optim.param_groups[0]["lr"]
    torch.manual_seed(1)
    tr2 = build_trainer(_trainer_cfg())
    assert not torch.equal(tr2.model_default.prompt_learner.ctx, tr.model_default.prompt_learner.ctx)
    assert tr2.resume_model_if_exist(str(tmp_path)) == 1
    assert torch.equal(tr2.model_default.prompt_learner.ctx, tr.model_default.prompt_learner.ctx)
    b1 = tr.optim.state_dict()["state"][0]["momentum_buffer"]
    b2 = tr2.optim.state_dict()["state"][0]["momentum_buffer"]
    assert torch.equal(b1, b2) and tr2.optim.param_groups[0]["lr"] == lr_saved and tr2.sched.last_epoch == tr.sched.last_epoch
    assert build_trainer(_trainer_cfg()).resume_model_if_exist(str(tmp_path / "nothing_here")) == 0


def test_restart_into_the_same_output_dir_continues(tmp_path):
    """dassl/engine/trainer.py:409-413: before_train always looks for a checkpoint in OUTPUT_DIR (RESUME only overrides the
    directory), so a job restarted with the same OUTPUT_DIR continues from the saved epoch instead of overwriting it from 0."""
    from leclip_amd.registry import build_trainer
    cfg = _trainer_cfg()
    cfg.OUTPUT_DIR = str(tmp_path / "run")
    tr = build_trainer(cfg)
    tr.before_train()
    assert tr.start_epoch == 0                      # nothing there yet: from scratch
    tr.save_model(0, tr.output_dir)
    tr2 = build_trainer(cfg)
    tr2.before_train()
    assert tr2.start_epoch == 1 and torch.equal(tr2.model_default.prompt_learner.ctx, tr.model_default.prompt_learner.ctx)
    other = _trainer_cfg()
    other.OUTPUT_DIR = str(tmp_path / "elsewhere")
    other.RESUME = str(tmp_path / "run")           # RESUME overrides where to look
    tr3 = build_trainer(other)
    tr3.before_train()
    assert tr3.start_epoch == 1


# ------------------------------------------------------------------------------ sharded evaluation inside the trainer plug-in
def _eval_batches():
    g = torch.Generator().manual_seed(11)
    batches = []
    for b, scales in ((5, (4, 9)), (3, (4, 9)), (1, (4, 9)), (4, (4, 9))):
        img = torch.randn(b, 3, 4, 4, generator=g)
        lab = (torch.rand(b, 80, generator=g) < 0.1).long()
        blocks = [torch.randn(b, w, 3, 4, 4, generator=g) for w in scales]
        batches.append({"img": img, "label": lab, "img_blocks": blocks})
    return batches


def _stub_inference(x, name):   # row-wise deterministic "scores" (global, local): what the batch-invariant HIP path guarantees
    f = x.flatten(1)
    base = torch.linspace(0.1, 1.0, 80)[None, :]
    return torch.sin(f.sum(1, keepdim=True) * base) * 0.6, torch.cos(f[:, :7].sum(1, keepdim=True) * base) * 0.6


def _torch_window_aggregate(global_logits, window_logits, threshold=0.3, weight=1.4):   # CDD.py:654-660 (the HIP kernel's arithmetic)
    alpha, beta = window_logits.max(dim=1)[0], window_logits.min(dim=1)[0]
    return weight * torch.where(alpha > threshold, alpha, beta) + global_logits


def _eval_trainer(with_windows):
    from leclip_amd.hip import ops
    from leclip_amd.registry import build_evaluator, build_trainer
    cfg = _trainer_cfg()
    cfg.merge_from_list(["DATALOADER.TEST.BATCH_SIZE", "3"])
    ops.window_aggregate = _torch_window_aggregate          # CPU test of the sharding logic: the aggregation kernel itself is GPU-tested
    batches = _eval_batches()
    if not with_windows:
        for bt in batches:
            del bt["img_blocks"]
    tr = build_trainer(cfg, evaluator=build_evaluator(cfg))
    tr.test_loader = batches
    tr.model_inference = _stub_inference
    return tr


def _eval_worker(rank, world, port, with_windows, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    calls = {"n": 0}
    real = dist.all_gather

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    dist.all_gather = counted
    tr = _eval_trainer(with_windows)
    calls["n"] = 0
    value = tr.test()
    out_q.put((rank, float(value), calls["n"], tr.evaluator.evaluate()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("with_windows", [False, True])
def test_trainer_test_is_sharded_with_one_collective_per_epoch(with_windows):
    """Caption_distill_double.test() under WORLD_SIZE = 2 (gloo): every rank scores its shard of each batch's images and of each
    scale's window list, ONE all-gather per epoch, and every rank's evaluator reports the single-process metric exactly - ragged
    batches, a batch with fewer images than ranks, global + local scores, sliding-window aggregation."""
    single = _eval_trainer(with_windows)
    want = float(single.test())
    want_all = single.evaluator.evaluate()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, with_windows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, value, n_collectives, all_metrics in res:
        assert n_collectives == 1, (rank, n_collectives)
        assert value == want and all_metrics == want_all
