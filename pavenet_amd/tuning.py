"""Per-shape selection of the vendor GEMM kernel (PyTorch TunableOp over hipBLASLt + rocBLAS).

The library's default heuristic leaves 5-15 % on some of this model's fp32 shapes (e.g. the FFN's
[625 044 x 1024] x [1024 x 256] GEMM: 2.60 ms by default, 2.22 ms = 147 TFLOP/s with the rocBLAS
solution TunableOp finds).  `pavenet_amd/data/tunableop_gfx950.csv` holds the selections measured
on an MI355X for the bench workload's shapes; they are applied WITHOUT tuning at run time (the
file's validators -- PyTorch / hipBLASLt / rocBLAS versions and gfx arch -- must match, otherwise
PyTorch ignores it and the defaults are used).  Shapes that are not in the file use the default
kernel; `use_tuned_gemms(tune=True, path=...)` measures new shapes and writes them to `path`."""
import os

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data',
                            'tunableop_gfx950.csv')


def use_tuned_gemms(path=None, tune=False, max_tuning_ms=100):
    """Enable TunableOp with the shipped (or given) selection file.  Returns the file used."""
    import torch.cuda.tunable as tn
    path = path or DEFAULT_FILE
    tn.enable(True)
    tn.tuning_enable(bool(tune))
    tn.set_filename(path, insert_device_ordinal=False)
    if tune:
        tn.set_max_tuning_duration(int(max_tuning_ms))
    elif os.path.exists(path):
        tn.read_file(path)
    return path


def disable():
    import torch.cuda.tunable as tn
    tn.enable(False)
