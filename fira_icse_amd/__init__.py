"""MI355X-native engine for the FIRA commit-message model (GNN encoder + dual-copy Transformer decoder).

The compute path is hand-written HIP for gfx950 behind a C ABI (``include/fira_hip.h``,
``fira_icse_amd/csrc``); this package is the host-side mirror of the reference's
Python surface (``run_model.py`` CLI, ``Dataset.py`` input format, ``Model.TransModel``).
"""
import os as _os

# hardware queues for the search's batches in flight (see run_model.py; effective only if the HIP runtime has not been
# initialised yet -- run_model.py and bench.py set it before importing torch)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import FiraConfig  # noqa: F401,E402

__version__ = "0.1.0"
