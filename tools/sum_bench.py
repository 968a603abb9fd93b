#!/usr/bin/env python3
"""One line per bench JSON file: step, topology-cached step, per-kernel table, training leg (same-box comparisons)."""
import json
import sys

for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line)
            tc = d.get("topology_cached") or {}
            print(f, round(d["ms_per_step"], 4), round(tc["ms_per_step"], 4) if tc else None,
                  {k: round(v, 4) for k, v in d.get("kernel_ms_per_step", {}).items()}, (d.get("train") or {}).get("ms_per_step"))
