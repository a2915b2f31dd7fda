#!/usr/bin/env python
"""Drop-in for the reference driver: ``python run_model.py train`` / ``python run_model.py test``
(reference run_model.py:417-425), running on the MI355X engine.

Kept from the reference: the positional ``train|test`` stage, every path relative to the working directory
(``DataSet/*.json``, ``VOCAB_UPPER_CASE``, ``all_index``, ``best_model.pt`` = ``torch.save(state_dict)`` with the
reference's 338 keys, ``OUTPUT/train_process``, ``OUTPUT/dev_output``, ``OUTPUT/output_fira``), the hyper-parameters of
its ``args`` dict as defaults, seed 0, the order in which the global RNGs are consumed (split shuffle, weight
initialisation, per-epoch DataLoader permutation), dev-BLEU checkpoint selection from epoch 15 every 10 batches, and
beam search with the reference's scoring quirks at test time.

New (all optional): overrides for what the reference hard-codes (``--batch-size``, ``--splits``, ``--epochs``,
``--beam``, ...), and multi-GPU through one process per GPU (``torchrun --nproc-per-node N run_model.py train``):
commits of the global batch are sharded over ranks and gradients all-reduced over RCCL, instead of the reference's
single-process ``nn.DataParallel``.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

# The search keeps several batches in flight, each on its own HIP stream (decode.Searcher.greedy_many); HIP multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two lanes that land on one queue run one after the other: four lanes
# took 0.38 ms per batch-step on 4 queues and 0.14 on 8 (profiles/r6_probes.md).  Read by the HIP runtime when it initialises,
# so it is set before torch is imported; training is unaffected (same-box triple).  An exported value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from fira_icse_amd import data, metrics, text                      # noqa: E402
from fira_icse_amd.config import EOS, PAD, START, UNK, FiraConfig                 # noqa: E402
from fira_icse_amd.parallel import gather_lines, init_from_env, shard_indices   # noqa: E402
from fira_icse_amd.prefetch import prefetch                        # noqa: E402


def seed_everything(seed=0):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def parse_args(argv):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("stage", choices=["train", "test"])
    ap.add_argument("--root", default=".", help="directory holding DataSet/, VOCAB_UPPER_CASE, all_index (default: cwd)")
    ap.add_argument("--splits", default=None, help="train,valid,test sizes (reference hard-codes 75000,8000,7661)")
    ap.add_argument("--batch-size", type=int, default=None, help="GLOBAL train batch (reference: 170 x n_gpu)")
    ap.add_argument("--test-batch-size", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=150)
    ap.add_argument("--beam", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--dev-from-epoch", type=int, default=15)
    ap.add_argument("--dev-every", type=int, default=10)
    ap.add_argument("--max-steps", type=int, default=0, help="stop after this many optimisation steps (0 = no limit)")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--loss-log", default=None, help="write one JSON line per optimisation step: the train-split positions "
                    "of the global batch and its mean token loss (synchronises every step, like run_model.py:112 does)")
    ap.add_argument("--save-optimizer", action="store_true", help="also write fira_train_state.pt (Adam moments, step)")
    ap.add_argument("--resume", action="store_true", help="start from best_model.pt (+ fira_train_state.pt if present)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="arithmetic of the nn.Linear products: f32 = "
                    "the reference's (fp32 MFMA, default); bf16 = BASELINE configs[2] (bf16 MFMA, fp32 accumulate; master "
                    "weights, LayerNorm, soft-max, loss and Adam stay fp32).  Applies to train and dev (teacher-forced BLEU); the test-time "
                    "SEARCH always runs the reference's fp32 arithmetic (decode.Searcher / fira_decode_step), whatever --dtype says")
    ap.add_argument("--grad-wire", choices=["auto", "f32", "bf16"], default="auto", help="multi-GPU, all-reduce path: wire format "
                    "of the gradient buckets; auto = bf16 with --dtype bf16 (half the bytes on xGMI; Adam, master weights and the "
                    "token normaliser stay fp32), else f32")
    ap.add_argument("--zero1", action="store_true", help="multi-GPU: reduce-scatter + Adam on the owned shard + all-gather "
                    "(Adam moments sharded over the ranks) instead of all-reduce + replicated Adam")
    return ap.parse_args(argv)


class Run:
    def __init__(self, a):
        self.a = a
        self.rank, self.world, self.local = init_from_env()
        torch.cuda.set_device(self.local)
        self.root = a.root
        with open(os.path.join(self.root, "DataSet", "word_vocab.json")) as f:
            self.vocab = json.load(f)
        with open(os.path.join(self.root, "DataSet", "ast_change_vocab.json")) as f:
            ast_vocab = json.load(f)
        self.r_vocab = {v: k for k, v in self.vocab.items()}
        # the kernels and the search loop use the special ids as constants (config.PAD/EOS/START/UNK); the reference looks
        # them up in word_vocab.json (run_model.py:205-222): refuse a vocabulary that numbers them differently
        for tok, want in (("<pad>", PAD), ("<eos>", EOS), ("<start>", START), ("<unkm>", UNK)):
            if self.vocab.get(tok) != want:
                raise ValueError("word_vocab.json maps %r to %r; this engine requires %d" % (tok, self.vocab.get(tok), want))
        with open(os.path.join(self.root, "DataSet", "variable.json")) as f:
            self.var_maps = json.load(f)
        bs = a.batch_size if a.batch_size else 170 * self.world
        self.cfg = FiraConfig(lr=a.lr, batch_size=bs, test_batch_size=a.test_batch_size, epoches=a.epochs,
                              beam_size=a.beam, vocab_size=len(self.vocab), ast_change_vocab_size=len(ast_vocab))
        splits = tuple(int(x) for x in a.splits.split(",")) if a.splits else None
        # rank 0 builds the caches (first run only: minutes of pure Python); the others poll the file system for the
        # finished, matching cache -- no collective, so no process-group timeout can fire during the build
        self.sets = {n: data.TransDataset(self.cfg, n, root=self.root, splits=splits, seed=a.seed, build=self.rank == 0)
                     for n in ("train", "valid", "test")}
        with open(os.path.join(self.root, "all_index")) as f:
            self.all_index = json.load(f)
        os.makedirs(os.path.join(self.root, "OUTPUT"), exist_ok=True)

    def out(self, name):
        return os.path.join(self.root, "OUTPUT", name)

    def device_batch(self, store, idx):
        from fira_icse_amd.model import DeviceBatch
        return DeviceBatch(store.batch(idx), self.cfg, self.model.device_)

    # ------------------------------------------------------------------------------ dev (run_model.py:118-184)
    @torch.no_grad()
    def dev(self, epoch):
        cfg, store = self.cfg, self.sets["valid"].store
        valid_index = self.all_index["valid"]
        self.model.eval()
        mine = shard_indices(list(range(len(store))), self.rank, self.world)
        lines, total = [], 0.0
        bs = max(1, cfg.batch_size // self.world)
        for lo in range(0, len(mine), bs):
            idx = mine[lo:lo + bs]
            ids = self.model.forward_dev(self.device_batch(store, idx)).cpu().tolist()
            for k, i in enumerate(idx):
                sen = text.dev_sentence(ids[k], store.sou[i], store.sub_token[i], cfg.vocab_size, cfg.sou_len, EOS)
                s = " ".join(self.r_vocab[t] for t in sen).replace("<pad>", "").replace("<unkm>", "\U0001F605").strip()
                hyp = s.split()
                ref_ids = store.tar[i].tolist()
                ref = [self.r_vocab[t] for t in ref_ids[1:ref_ids.index(EOS)]]
                b = metrics.sentence_bleu_method2([ref], hyp)
                total += b
                back = {v: k2 for k2, v in self.var_maps[valid_index[i]].items()}
                lines.append(" ".join(back.get(t, t) for t in hyp) + "," + str(b))
        if self.world > 1:
            t = torch.tensor([total], dtype=torch.float64, device=self.model.device_)
            torch.distributed.all_reduce(t)
            total = float(t.item())
            lines = gather_lines(lines)
        self.model.train()
        return total / max(1, len(store)), "\n".join(lines) + "\n"

    # ------------------------------------------------------------------------------ train (run_model.py:83-117,382-399)
    def train(self):
        from fira_icse_amd.model import TransModel
        from fira_icse_amd.train import Trainer
        a, cfg = self.a, self.cfg
        store = self.sets["train"].store
        self.model = TransModel(cfg, device="cuda:%d" % self.local)       # consumes the torch RNG like the reference
        if a.resume and os.path.exists(os.path.join(self.root, "best_model.pt")):
            self.model.load_state_dict(torch.load(os.path.join(self.root, "best_model.pt"), map_location="cpu"))
        self.model.compute_dtype = a.dtype
        self.model.set_dropout_stream(a.seed, self.rank)           # masks depend on (--seed, rank, step)
        wire = a.grad_wire if a.grad_wire != "auto" else ("bf16" if a.dtype == "bf16" else "f32")
        trainer = Trainer(self.model, lr=cfg.lr, distributed=self.world > 1, zero1=a.zero1, grad_wire=wire)
        state_path = os.path.join(self.root, "fira_train_state.pt")
        if a.resume and os.path.exists(state_path):
            trainer.load_state_dict(torch.load(state_path, map_location=self.model.device_))
        self.model.train(not a.no_dropout)
        best_bleu, steps = -1.0, 0
        n_batches = -(-len(store) // cfg.batch_size)
        for epoch in range(cfg.epoches):
            total_data, t0 = 0, time.time()
            def prepare(gidx):                                     # worker thread: collate + H2D of this rank's shard
                mine = shard_indices(gidx, self.rank, self.world)   # DataParallel.scatter's contiguous chunks
                return gidx, (self.device_batch(store, mine) if mine else None)

            batches = prefetch(data.iterate_batches(len(store), cfg.batch_size, shuffle=True), prepare, depth=2,
                               device=self.model.device_)
            for idx_b, (gidx, db) in enumerate(batches):
                if epoch >= a.dev_from_epoch and idx_b % a.dev_every == 0:
                    cur_bleu, output_str = self.dev(epoch)
                    if self.rank == 0:
                        with open(self.out("train_process"), "a") as f:
                            f.write("epoch: {} batch: {} dev bleu: {} is better: {}\n".format(
                                epoch, idx_b, cur_bleu, cur_bleu > best_bleu))
                    if cur_bleu > best_bleu:
                        best_bleu = cur_bleu
                        opt_state = trainer.state_dict() if a.save_optimizer else None     # collective with --zero1
                        if self.rank == 0:
                            torch.save(self.model.state_dict(), os.path.join(self.root, "best_model.pt"))
                            if a.save_optimizer:
                                torch.save(opt_state, state_path)
                            with open(self.out("dev_output"), "w") as f:
                                f.write(output_str)
                    self.model.train(not a.no_dropout)
                trainer.step(db)                     # db None (empty shard of a short tail batch): still joins the collectives
                total_data += len(gidx)
                steps += 1
                if a.loss_log and self.rank == 0:
                    with open(a.loss_log, "a") as f:
                        f.write(json.dumps({"epoch": epoch, "batch": idx_b, "index": [int(i) for i in gidx],
                                            "loss": trainer.last_loss()}) + "\n")
                if idx_b % 10 == 0 and self.rank == 0:
                    print("epoch: %d batch: %d/%d  data: %d/%d loss: %.4f  (%.1f commits/s)" % (
                        epoch, idx_b, n_batches, total_data, len(store), trainer.last_loss(),
                        total_data / max(time.time() - t0, 1e-9)), flush=True)
                if a.max_steps and steps >= a.max_steps:
                    break
            batches.close()                          # stops the worker thread and drops its prepared batches
            if a.max_steps and steps >= a.max_steps:
                break
        if best_bleu < 0:                               # never reached a dev point (short runs): keep the last weights
            opt_state = trainer.state_dict() if a.save_optimizer else None                 # collective with --zero1
            if self.rank == 0:
                torch.save(self.model.state_dict(), os.path.join(self.root, "best_model.pt"))
                if a.save_optimizer:                     # ... and the optimizer state that belongs to them (--resume)
                    torch.save(opt_state, state_path)
        return best_bleu

    # ------------------------------------------------------------------------------ test (run_model.py:187-380,401-415)
    @torch.no_grad()
    def test(self):
        from fira_icse_amd.model import TransModel
        from fira_icse_amd.decode import Searcher
        cfg, store = self.cfg, self.sets["test"].store
        test_index = self.all_index["test"]
        self.model = TransModel(cfg, device="cuda:%d" % self.local, init=False)
        self.model.load_state_dict(torch.load(os.path.join(self.root, "best_model.pt"), map_location="cpu"))
        self.model.compute_dtype = self.a.dtype
        self.model.eval()
        search = Searcher(self.model)
        mine = shard_indices(list(range(len(store))), self.rank, self.world)
        lines, n_tok, t0 = [], 0, time.time()
        # greedy: groups of `in_flight` batches share the GPU (independent launch chains: decode.Searcher.greedy_many; four lanes
        # on eight hardware queues: 0.14 ms per batch-step against 0.33 one at a time); the output order stays
        # all_index['test'] order (run_model.py:372)
        group = int(os.environ.get("FIRA_DECODE_IN_FLIGHT", "4")) if cfg.beam_size == 1 else 1
        starts = list(range(0, len(mine), cfg.test_batch_size))
        for g0 in range(0, len(starts), group):
            idxs = [mine[lo:lo + cfg.test_batch_size] for lo in starts[g0:g0 + group]]
            dbs = [self.device_batch(store, idx) for idx in idxs]
            if cfg.beam_size == 1:
                outs = [search.best(*r) for r in search.greedy_many(dbs, in_flight=group)]
            else:
                outs = [search.best(*search.beam(dbs[0], cfg.beam_size))]
            for idx, hyps in zip(idxs, outs):
                for h, i in zip(hyps, idx):
                    lines.append(text.detokenize(h, self.r_vocab, self.var_maps[test_index[i]]))
                    n_tok += max(len(h) - 1, 0)
            if self.rank == 0:
                done = min(len(mine), starts[min(g0 + group, len(starts)) - 1] + cfg.test_batch_size)
                print("data: %d/%d  (%.1f tokens/s)" % (done, len(mine), n_tok / max(time.time() - t0, 1e-9)), flush=True)
        lines = gather_lines(lines)
        if self.rank == 0:
            with open(self.out("output_fira"), "w") as f:
                f.write("".join(l + "\n" for l in lines))
        return lines


def main(argv=None):
    a = parse_args(sys.argv[1:] if argv is None else argv)
    seed_everything(a.seed)
    run = Run(a)
    if a.stage == "train":
        run.train()
    else:
        run.test()
    if run.world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
