"""MI355X-native engine for the FIRA commit-message model (GNN encoder + dual-copy Transformer decoder).

The compute path is hand-written HIP for gfx950 behind a C ABI (``include/fira_hip.h``,
``fira_icse_amd/csrc``); this package is the host-side mirror of the reference's
Python surface (``run_model.py`` CLI, ``Dataset.py`` input format, ``Model.TransModel``).
"""
from .config import FiraConfig  # noqa: F401

__version__ = "0.1.0"
