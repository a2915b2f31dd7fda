"""In-kernel wall-clock stamps of gcn_fused_kernel (debug build only: GF_STAMP): phases of the last forward / backward launch."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from fira_icse_amd import data, synth
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch
from fira_icse_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = FiraConfig()
store = data.process_raw(cfg, synth.generate_dataset(2 * B, seed=1000))
torch.manual_seed(0)
model = TransModel(cfg, device="cuda"); model.train()
tr = Trainer(model)
dbs = [DeviceBatch(store.batch(list(range(B * i, B * i + B))), cfg, model.device_) for i in range(2)]
for i in range(6):
    tr.step(dbs[i % 2])
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(R, "fira_icse_amd", "libfira_hip.so"))
buf = (ctypes.c_ulonglong * (2 * 256 * 8))()
lib.fira_gf_dbg_read(buf, 2 * 256 * 8)
a = np.array(buf[:], dtype=np.float64).reshape(2, 256, 8)
for d, nm in ((0, "forward"), (1, "backward")):
    x = a[d]
    t0 = x[:, 0].min()
    ph = np.diff(x[:, :6], axis=1) / 100.0
    names = ["gather (+first weight request)", "barrier 1 wait", "product", "barrier 2 wait + acc->LDS + barrier 3", "row phase"]
    print(nm, "start spread %.1f us, end max %.1f us" % ((x[:, 0].max() - t0) / 100.0, (x[:, 5].max() - t0) / 100.0))
    for k, n in enumerate(names):
        print("   %-40s min %.1f  med %.1f  max %.1f us" % (n, ph[:, k].min(), np.median(ph[:, k]), ph[:, k].max()))
