"""Data-parallel logic on CPU with the gloo backend (world_size 2): sharding == DataParallel.scatter chunking, and the
2-rank reduced gradient with the GLOBAL token normaliser == the single-process gradient of the same global batch
(SURVEY.md §8d config 3 equivalence check).  The gradient producer here is the CPU oracle (tests only)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from fira_icse_amd import data
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.parallel import GradReducer, ShardedOptimizerComm, gather_lines, shard_indices, shard_range


def test_shard_range_is_dataparallel_scatter_chunking():
    for n in (1, 5, 8, 170, 680, 31):
        for world in (1, 2, 3, 4, 8):
            chunks = [c.tolist() for c in torch.arange(n).chunk(world)]
            chunks += [[]] * (world - len(chunks))
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert list(range(lo, hi)) == chunks[r], (n, world, r)
            assert sum((shard_indices(list(range(n)), r, world) for r in range(world)), []) == list(range(n))


def _flat_grad(P, layout):
    flat = torch.zeros(layout.total)
    views = layout.views(flat)
    for k, p in P.items():
        if p.grad is not None:
            views[k].copy_(p.grad)
    return flat


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, util.REPO)
    torch.set_num_threads(4)
    from oracle import fira_oracle as O
    from fira_icse_amd.model import ParamLayout, reference_init_state_dict
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = FiraConfig()
    store = data.process_raw(cfg, util.load_golden_raw())
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)["train"][:4]
    torch.manual_seed(0)
    sd = reference_init_state_dict(cfg)
    layout = ParamLayout(cfg)

    def grads_of(ids):
        tb = util.to_torch_batch(store.batch(ids), cfg)
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ls, nt = O.forward(P, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                           tb["sub_token"], "train")
        ls.backward()                                   # gradient of loss_SUM, like the engine produces
        return _flat_grad(P, layout), float(ls.detach()), int(nt)

    mine = shard_indices(idx, rank, world)
    g, ls, nt = grads_of(mine)
    red = GradReducer(layout.split, layout.live)
    stats = torch.tensor([ls, float(nt)])
    red.finish(g, stats)
    lines = gather_lines(["r%d-%d" % (rank, i) for i in mine])
    if rank == 0:
        gf, lsf, ntf = grads_of(idx)
        torch.save(dict(g=g / stats[1], gf=gf / ntf, stats=stats, full=(lsf, ntf), lines=lines, idx=idx), tmp)
    dist.barrier()
    dist.destroy_process_group()


def _worker_empty(rank, world, port, tmp):
    """Rank 1 holds an EMPTY shard (global batch of one commit over two ranks): it contributes a zero gradient and
    zero (loss_sum, n_tok) to the same collectives, as Trainer.step(None) does on the GPU."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, util.REPO)
    torch.set_num_threads(4)
    from oracle import fira_oracle as O
    from fira_icse_amd.model import ParamLayout, reference_init_state_dict
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = FiraConfig()
    store = data.process_raw(cfg, util.load_golden_raw())
    idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)["train"][:1]
    torch.manual_seed(0)
    sd = reference_init_state_dict(cfg)
    layout = ParamLayout(cfg)
    mine = shard_indices(idx, rank, world)
    assert (len(mine) == 1) == (rank == 0)
    if mine:
        tb = util.to_torch_batch(store.batch(mine), cfg)
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ls, nt = O.forward(P, cfg, tb["sou"], tb["tar"], tb["mark"], tb["ast_change"], tb["edge"], tb["tar_label"],
                           tb["sub_token"], "train")
        ls.backward()
        g, stats = _flat_grad(P, layout), torch.tensor([float(ls.detach()), float(nt)])
    else:
        g, stats = torch.zeros(layout.total), torch.zeros(2)
    ref = g.clone()
    red = GradReducer(layout.split, layout.live)
    red.finish(g, stats)
    out = [None, None]
    dist.all_gather_object(out, (g[:layout.live].clone(), stats.clone()))
    if rank == 0:
        torch.save(dict(g0=out[0][0], g1=out[1][0], s0=out[0][1], s1=out[1][1], ref=ref[:layout.live]), tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_rank_with_empty_shard_joins_the_collectives(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 29300 + (os.getpid() % 200)
    mp.spawn(_worker_empty, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    assert torch.equal(r["g0"], r["g1"]) and torch.equal(r["g0"], r["ref"])      # both hold rank 0's gradient
    assert torch.equal(r["s0"], r["s1"]) and float(r["s0"][1]) > 0               # and the global token count


def test_two_rank_gradient_equals_single_rank(tmp_path):
    out = str(tmp_path / "res.pt")
    port = 29600 + (os.getpid() % 200)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    assert int(r["stats"][1]) == r["full"][1]                                   # global token count
    assert abs(float(r["stats"][0]) - r["full"][0]) <= 1e-5 * r["full"][0]      # global loss sum
    err = float((r["g"] - r["gf"]).double().norm() / r["gf"].double().norm())
    assert err < 1e-5, err
    assert r["lines"] == ["r0-%d" % i for i in r["idx"][:2]] + ["r1-%d" % i for i in r["idx"][2:]]   # ordered gather


def _zero_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    split, live, total = 1003, 2411, 2600                    # ragged on purpose: neither bucket divides by the world size
    z = ShardedOptimizerComm(split, live, total)
    g = torch.Generator().manual_seed(100 + rank)
    gbuf = torch.randn(total, generator=g)
    flat = torch.arange(total, dtype=torch.float32)          # identical "parameters" on every rank
    want_sum = sum(torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    covered = torch.zeros(total)
    for b in (0, 1):
        q = z.buckets[b]
        out = torch.zeros(q["chunk"])
        z.reduce_scatter(b, gbuf, out)
        lo, hi = z.owned(b)
        assert torch.allclose(out[:hi - lo], want_sum[lo:hi], atol=1e-6)           # the owned shard of the summed gradient
        flat[lo:hi] -= 0.5 * out[:hi - lo]                                            # "optimizer" on the owned shard only
        covered[lo:hi] += 1
        z.all_gather(b, flat)
    dist.all_reduce(covered)
    assert torch.equal(covered[:live], torch.ones(live)) and float(covered[live:].sum()) == 0     # a partition of [0, live)
    want = torch.arange(total, dtype=torch.float32)
    want[:live] -= 0.5 * want_sum[:live]
    assert torch.allclose(flat, want, atol=1e-6)              # every rank holds every updated parameter; the tail is untouched
    shards = [torch.full((q["chunk"],), float(rank + 1)) for q in z.buckets]
    full = z.gather_full(shards, total)                       # checkpoint view: rank r's value on rank r's ranges
    for r in range(world):
        for q in z.buckets:
            a = min(q["hi"], q["lo"] + r * q["chunk"])
            assert torch.all(full[a:min(q["hi"], a + q["chunk"])] == r + 1)
    assert float(full[live:].abs().sum()) == 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_comm_partitions_reduces_and_gathers(tmp_path, world):
    """ZeRO-1 comm (reduce-scatter / owned shard / all-gather) over ragged buckets, gloo, world_size 2 and 3."""
    mp.spawn(_zero_worker, args=(world, 29650 + (os.getpid() % 200) + world, str(tmp_path)), nprocs=world, join=True)


def _wire_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    split, live, total = 1024, 4096, 4160                    # (bucket bounds are multiples of 64 elements: csrc/layout.cpp)
    mk = lambda r: torch.randn(total, generator=torch.Generator().manual_seed(7 + r)) * torch.logspace(-3, 1, total)
    res = {}
    for wire in ("f32", "bf16"):
        g = mk(rank)
        stats = torch.tensor([3.5 + rank, 11.0 + rank])
        red = GradReducer(split, live, wire=wire)
        red.finish(g, stats)
        res[wire] = (g.clone(), stats.clone(), red.bytes_per_step())
    if rank == 0:
        res["want"] = sum(mk(r) for r in range(world))
        torch.save(res, tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_gradient_wire_format(tmp_path):
    """GradReducer(wire="bf16") (BASELINE configs[2]; SURVEY.md 2.2: 55.6 MB instead of 111.2 MB per step): both buckets are
    rounded to bf16, summed over the ranks in bf16 and widened back -- the stats pair and everything outside [0, live) stay
    fp32.  Two ranks: rel-L2 of the reduced gradient against the fp32 reduction <= 4e-3 (one rounding per contribution and
    one per partial sum, 2^-9 each)."""
    out = str(tmp_path / "res.pt")
    port = 29400 + (os.getpid() % 200)
    mp.spawn(_wire_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    live = 4096
    g32, s32, b32 = r["f32"]
    g16, s16, b16 = r["bf16"]
    assert torch.allclose(g32[:live], r["want"][:live], rtol=1e-6, atol=1e-7)
    assert torch.equal(s32, s16) and torch.equal(s32, torch.tensor([8.0, 23.0]))         # the normaliser never goes bf16
    err = float((g16[:live] - g32[:live]).double().norm() / g32[:live].double().norm())
    assert 1e-5 < err < 4e-3, err
    assert torch.equal(g16[live:], g32[live:])                                             # dead tensors: untouched
    assert b32 == 4 * live + 8 and b16 == 2 * live + 8
