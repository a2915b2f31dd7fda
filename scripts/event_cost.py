"""What a stream fork (hipEventRecord on the chain's stream + hipStreamWaitEvent on another) costs a dependent chain of
small kernels: n = 2000 tiny kernels enqueued by ONE C call (no Python per launch), GPU time from events."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fira_icse_amd import _lib

lib = _lib.lib()
scratch = torch.zeros(256, device="cuda")
n = 2000
names = {0: "plain chain", 1: "fork after every kernel", 2: "event record after every kernel", 3: "fork every 8th kernel",
         4: "stop event on every kernel", 5: "stop-event fork after every kernel", 6: "stop-event fork every 8th kernel"}
for mode in (0, 1, 2, 3, 4, 5, 6, 0, 10, 11, 12, 13, 14, 15, 16):
    ahead = mode >= 10          # host far ahead of the GPU: a 30 ms spin kernel first, so the chain is queued before it starts
    mode = mode % 10
    for rep in range(2):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if ahead:
            torch.cuda._sleep(60_000_000)
        a.record()
        _lib.check(lib.fira_debug_chain(_lib.cur_stream(), n, mode, _lib.ptr(scratch)), "debug_chain")
        b.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
    print("%-34s %s %.2f us per kernel on the GPU, host enqueue %.2f us per kernel" % (names[mode], "[queued ahead]" if ahead else "", a.elapsed_time(b) * 1e3 / n, t_host * 1e6 / n))
