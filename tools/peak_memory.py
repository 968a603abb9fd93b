#!/usr/bin/env python3
"""Peak device memory of a bench.py run: `python tools/peak_memory.py --workload c5 --mode train --steps 3 ...` (bench.py's
own arguments) prints torch's max allocated / reserved bytes after the run."""
import contextlib
import io
import os
import runpy
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path(sys.argv[0], run_name="__main__")
print(" ".join(sys.argv[1:]), "| max allocated GB", round(torch.cuda.max_memory_allocated() / 1e9, 1), "| max reserved GB",
      round(torch.cuda.max_memory_reserved() / 1e9, 1))
