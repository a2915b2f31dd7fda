"""bench.py's stand-alone aggregation legs alone (spmm_b64 / spmm_b64_compact / spmm_cfg5 / gcn_cfg5)."""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
cfg = FiraConfig()
store = data.process_raw(cfg, synth.generate_dataset(256, seed=1000))
out = bench.spmm_standalone(cfg, store)
for k in ("spmm_b64", "spmm_b64_compact"):
    print(k, "%.2f us  frac %.3f" % (out[k]["avg_launch_us"], out[k]["frac"]))
c5 = out["spmm_cfg5"]
print("spmm_cfg5", {q: round(c5[q]["avg_launch_us"], 1) for q in ("v1", "v2", "v3", "v4")}, "frac %.3f  frac_bf16 %.3f" % (c5["frac"], c5["frac_bf16"]))
g5 = out["gcn_cfg5"]
for n in ("f32", "bf16"):
    print("gcn_cfg5", n, {q: round(v, 1) if isinstance(v, float) else v for q, v in g5[n].items()})
