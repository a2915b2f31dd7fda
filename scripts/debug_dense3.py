import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
import ctypes as C
from fira_icse_amd import data, _lib
from fira_icse_amd.config import FiraConfig
from fira_icse_amd.model import TransModel, DeviceBatch, reference_init_state_dict, _as_tensor
cfg = FiraConfig()
store = data.process_raw(cfg, util.load_golden_raw())
idx = data.split_index(*util.GOLDEN_SPLIT, seed=0)
hb = store.batch(idx["train"][:util.GOLDEN_B])
torch.manual_seed(0)
sd = util.perturb_state_dict(reference_init_state_dict(cfg), seed=1)
model = TransModel(cfg, init=False)
model.load_state_dict(sd)
model.eval()
dense = DeviceBatch(hb, cfg, skip_padding=(os.environ.get("DBG_SKIP", "0") == "1"))
lib = _lib.lib()
n = lib.fira_decode_workspace_bytes(C.byref(model.dims), dense.B, 1)
ws = torch.empty(n, dtype=torch.uint8, device="cuda")
_lib.check(lib.fira_decode_begin(_lib.cur_stream(), C.byref(model.dims), C.byref(dense.struct), _lib.ptr(model.flat.data), _lib.ptr(ws), ws.numel(), 1), "x")
ptr = lib.fira_decode_memory(C.byref(model.dims), _lib.ptr(ws), dense.B, 1)
mem = _as_tensor(ptr, (dense.B * cfg.mem_len, 256), "cuda").clone().cpu()
tag = sys.argv[1]
torch.save(mem, "/tmp/mem_%s.pt" % tag)
print(tag, "finite", bool(torch.isfinite(mem).all()), "absmax", float(mem.abs().max()))
if len(sys.argv) > 2:
    o = torch.load("/tmp/mem_%s.pt" % sys.argv[2])
    d = (mem - o).abs().max(1).values
    sou = torch.from_numpy(hb.sou); sub = torch.from_numpy(hb.sub_token)
    valid = torch.cat([sou != 0, sub != 0], 1).reshape(-1)
    print("rows differing:", int((d > 0).sum()), "valid among them:", int(((d > 0) & valid).sum()), "max diff", float(d.max()))
    bad = torch.nonzero(d > 0).flatten()[:10]
    print("first differing rows (b, slot):", [(int(r) // cfg.mem_len, int(r) % cfg.mem_len) for r in bad])
