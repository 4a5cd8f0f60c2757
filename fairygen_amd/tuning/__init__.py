"""hipBLASLt solution table for the DiT GEMM shapes whose default heuristic pick is poor.

The GEMMs of the hot path stay on the library (``F.linear`` -> hipBLASLt), as ``north_star`` leaves them.  At the token
counts of the multi-GPU layouts (N/2, N/4, N/8 rows per rank) the heuristic's choice for ffn.2 (K = 14336, 3072 output
columns: few output tiles, long K) runs at 0.75 PFLOP/s where another solution of the same library reaches 1.15-1.35.
``tools/tune_gemms.py`` searches all solutions once on an MI355X (PyTorch TunableOp) and keeps only the shapes with a
clear gain in ``gfx950_gemm.csv``; this module loads that table READ-ONLY (no tuning at run time) and ``linear``
switches TunableOp on just around the one call, so every other GEMM keeps the default path.  A table written by another
hipBLASLt / PyTorch build fails TunableOp's validator check and is ignored (default solutions, same results).
"""
import os

import torch
import torch.nn.functional as F

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gfx950_gemm.csv")
_state = {"loaded": None}


def load_table():
    """Read the committed table once; returns True when its entries are active."""
    if _state["loaded"] is None:
        ok = False
        if torch.cuda.is_available() and os.path.exists(TABLE) and os.environ.get("FAIRYGEN_GEMM_TABLE", "1") != "0":
            import torch.cuda.tunable as tunable
            tunable.tuning_enable(False)
            tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fairygen_tunableop_{os.getpid()}.csv"))
            tunable.enable(True)
            ok = bool(tunable.read_file(TABLE))
            tunable.enable(False)
        _state["loaded"] = ok
    return _state["loaded"]


def linear(x, weight, bias):
    """F.linear with the table's hipBLASLt solution for this shape when it has one (else the default heuristic)."""
    if not load_table():
        return F.linear(x, weight, bias)
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    try:
        return F.linear(x, weight, bias)
    finally:
        tunable.enable(False)
