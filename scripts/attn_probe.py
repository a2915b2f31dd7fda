"""Stand-alone timing of the cross-attention kernels at the engine's shapes.

    python scripts/attn_probe.py [batch] [pitch_floats] [layout]

One launch = B commits x 8 heads, ragged query rows (about half of the 30 target positions per commit), 370 memory slots of
which a FIRA-shaped prefix of the code tokens (<= 210) and of the sub-tokens (<= 160) is valid.
layout:
    dense   K | V rows inside a [B*370, pitch] buffer, padded slots masked (the round-3 engine layout)
    ragged  the valid rows only, commit after commit ([n_valid, pitch]), key ranges through k_off (round 4)
    self    the decoder's causal self-attention: q | k | v are column slices of one [R, 768] buffer of the compact target rows
pitch 3136 = 6 layers x 512 + 64 pad (the engine's); 3072 = unpadded; 512 = one buffer per layer.
Rotates over enough buffers to leave the last-level cache, times forward and backward with HIP events, and prints the
microseconds per launch together with the bytes and the MFMA work a launch has, so that the number can be put against a roof:
    bytes  = K and V head slices of the valid keys (+ dK, dV in the backward) + Q / O / dO rows
    flop   = 2 * 32 * 32 * 32 per (live key tile, chain): 2 chains forward, 7 backward
Run it under `rocprofv3 --pmc ...` (scripts/pmc_traffic.sh shows the counter passes) to see where a workgroup waits.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import ops  # noqa: E402


def self_probe(B, dev, q_off, R):
    T, H = 30, 8
    qo = torch.from_numpy(q_off).to(dev)
    valid = torch.ones(B * T, dtype=torch.int32, device=dev)
    bufs = [torch.randn(R, 768, device=dev) * 0.5 for _ in range(8)]
    do = torch.randn(R, 256, device=dev) * 0.1
    tiles = B * H
    for name, bwd, chains in (("forward", False, 2), ("forward + backward", True, 9)):
        def run(i):
            x = bufs[i % 8]
            q, k, v = x[:, :256], x[:, 256:512], x[:, 512:]
            o = ops.attention_ragged_fwd(q, k, v, valid, qo, T, T, causal=True, self_kv=True)
            if bwd:
                ops.attention_ragged_bwd(q, k, v, valid, o, do, qo, T, T, causal=True, self_kv=True)
        def wrapper_only():
            torch.zeros_like(do)
            if bwd:
                torch.full_like(bufs[0][:, :256], 0.0); torch.full_like(bufs[0][:, :256], 0.0); torch.full_like(bufs[0][:, :256], 0.0)
        for i in range(6):
            run(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        a.record()
        for i in range(n):
            run(i)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / n
        a.record()
        for i in range(n):
            wrapper_only()
        b.record()
        torch.cuda.synchronize()
        us_w = a.elapsed_time(b) * 1e3 / n
        flop = tiles * chains * 2 * 32 * 32 * 32
        print("%-20s self batch %d: %.1f us per launch (%.1f incl. the wrapper's fills), %d rows, %.3f GFLOP" %
              (name, B, max(us - us_w, 1e-3), us, R, flop / 1e9))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    pitch = int(sys.argv[2]) if len(sys.argv) > 2 else 3136
    layout = sys.argv[3] if len(sys.argv) > 3 else "ragged"
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    T, S, H, D = 30, 370, 8, 256
    n_code = rng.integers(20, 200, size=B)
    n_sub = rng.integers(5, 120, size=B)
    valid = np.zeros((B, S), np.int32)
    for b in range(B):
        valid[b, :n_code[b]] = 1
        valid[b, 210:210 + n_sub[b]] = 1
    tq = rng.integers(6, 26, size=B)
    q_off = np.zeros(B + 1, np.int32)
    q_off[1:] = np.cumsum(tq)
    R = int(q_off[-1])
    n_valid = int(valid.sum())
    if layout == "self":
        return self_probe(B, dev, q_off, R)
    if layout == "ragged":
        k_off_np = np.zeros(B + 1, np.int32)
        k_off_np[1:] = np.cumsum(valid.sum(1))
        live_tiles = int(sum(-(-int(n) // 32) for n in valid.sum(1)))
        rows = n_valid
    else:
        live_tiles = sum(int(np.any(valid[b, 32 * t:32 * t + 32])) for b in range(B) for t in range(12))
        rows = B * S
    kv_bytes = n_valid * H * 128 * 2
    row_bytes = R * D * 4
    fwd_bytes, bwd_bytes = kv_bytes + 2 * row_bytes, 2 * kv_bytes + 4 * row_bytes
    chain_flop = 2 * 32 * 32 * 32
    fwd_flop, bwd_flop = live_tiles * H * 2 * chain_flop, live_tiles * H * 7 * chain_flop

    n_lay = max(1, min(6, pitch // 512))
    n_buf = max(2, int(600e6 // (rows * pitch * 4 * 2)) + 1)      # > the 256 MB last-level cache
    bufs = []
    for _ in range(n_buf):
        kv = torch.randn(rows, pitch, device=dev) * 0.5
        dkv = torch.zeros(rows, pitch, device=dev)
        bufs.append((kv, dkv))
    q = torch.randn(R, D, device=dev) * 0.5
    do = torch.randn(R, D, device=dev) * 0.1
    qo = torch.from_numpy(q_off).to(dev)
    if layout == "ragged":
        kvalid = torch.ones(n_valid, dtype=torch.int32, device=dev)
        k_off = torch.from_numpy(k_off_np).to(dev)
    else:
        kvalid = torch.from_numpy(valid).to(dev)
        k_off = None
    o_buf = torch.zeros(R, D, device=dev)

    def run(bwd, layer, i):
        kv, dkv = bufs[i % n_buf]
        k, v = kv[:, layer * 512:layer * 512 + 256], kv[:, layer * 512 + 256:layer * 512 + 512]
        o = ops.attention_ragged_fwd(q, k, v, kvalid, qo, T, S, k_off=k_off)
        if bwd:
            ops.attention_ragged_bwd(q, k, v, kvalid, o, do, qo, T, S, k_off=k_off)

    # the wrappers allocate + fill their outputs (torch kernels): time those alone and subtract
    def wrapper_only(bwd):
        kv, _ = bufs[0]
        k = kv[:, :256]
        torch.zeros_like(q)
        if bwd:
            torch.full_like(q, 0.0); torch.full_like(k, 0.0); torch.full_like(k, 0.0)

    for name, bwd, nbytes, flop in (("forward", False, fwd_bytes, fwd_flop), ("forward + backward", True, fwd_bytes + bwd_bytes, fwd_flop + bwd_flop)):
        for i in range(6):
            run(bwd, i % n_lay, i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 60
        a.record()
        for i in range(n):
            run(bwd, i % n_lay, i)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / n
        a.record()
        for i in range(n):
            wrapper_only(bwd)
        b.record()
        torch.cuda.synchronize()
        us_w = a.elapsed_time(b) * 1e3 / n
        k_us = max(us - us_w, 1e-3)
        print("%-20s %s batch %d pitch %d: %.1f us per launch (%.1f incl. the wrapper's fills), %d rows, %d valid keys, %d/%d live key "
              "tiles, %.1f MB -> %.2f TB/s, %.2f GFLOP -> %.1f TF/s" % (name, layout, B, pitch, k_us, us, R, n_valid, live_tiles, B * 12,
                                                                        nbytes / 1e6, nbytes / k_us / 1e6, flop / 1e9, flop / k_us / 1e6))


if __name__ == "__main__":
    main()
