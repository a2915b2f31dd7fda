"""What a stream fork (hipEventRecord on the chain's stream + hipStreamWaitEvent on another) costs a dependent chain of
small kernels: n = 2000 tiny kernels enqueued by ONE C call (no Python per launch), GPU time from events."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import _lib

lib = _lib.lib()
scratch = torch.zeros(256, device="cuda")
n = 2000
names = {0: "plain chain", 1: "fork after every kernel", 2: "event record after every kernel", 3: "fork every 8th kernel"}
for mode in (0, 1, 2, 3, 0):
    for rep in range(2):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        _lib.check(lib.fira_debug_chain(_lib.cur_stream(), n, mode, _lib.ptr(scratch)), "debug_chain")
        b.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
    print("%-34s %.2f us per kernel on the GPU, host enqueue %.2f us per kernel" % (names[mode], a.elapsed_time(b) * 1e3 / n, t_host * 1e6 / n))
