#!/usr/bin/env python3
"""Condense rocprofv3 counter-collection CSVs (one directory per --pmc pass of the SAME command) into one small JSON
that can be kept under profiles/.

    python tools/pmc_summary.py OUT.json --workload c3 PASS_DIR [PASS_DIR ...]

Per kernel of this library (ATen kernels are dropped): dispatches, average duration, registers / LDS / scratch as the
profiler reports them, the average of every collected counter per dispatch, and derived figures:

  hbm_bytes_per_launch   2 * FETCH_SIZE + WRITE_SIZE, KB -> bytes.  FETCH_SIZE is doubled because on gfx950 it
                         tallies a 128-B request of a wide streaming read as 64 B (/opt/skills/guides/
                         MI355X_MICROARCH.md, "HBM"); FETCH_SIZE and WRITE_SIZE come from separate passes (both do not
                         fit the TCC's four slots).
  valu_per_mfma          (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA: vector instructions issued next to each matrix
                         instruction (fp32 MFMA and VALU time add up on this part)
  wait_any_frac          SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of wave-cycles parked on s_waitcnt / barriers
  mfma_busy_frac         SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * SQ_BUSY_CYCLES-derived kernel cycles) is NOT formed here
                         (the two counters' units differ per block); instead
  mfma_pipe_util         SQ_INSTS_MFMA * cycles_per_instruction / (duration * 2.4 GHz * 1024 SIMDs), with 64 cycles for
                         v_mfma_f32_32x32x2_f32 and 32 for v_mfma_f32_16x16x4_f32 (the 16-row kernels)
"""
import argparse
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

OURS = ("mlp_", "scatter_sum", "gather_rows", "csr_", "xty_", "colsum", "adam_", "edge_features", "agg_fixup", "permute_index",
        "key_prep", "rowptr")


def short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:  # cut at the parameter list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()[:160]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--workload", default="")
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    csv.field_size_limit(1 << 30)
    per = defaultdict(lambda: {"counters": defaultdict(list), "dur": [], "meta": None})
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    if not any(s in k for s in OURS):
                        continue
                    e = per[k]
                    e["counters"][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    did = (f, r["Dispatch_Id"])
                    if did not in seen:
                        seen.add(did)
                        e["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                    e["meta"] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                                 "lds_bytes": int(r["LDS_Block_Size"]), "scratch_bytes": int(r["Scratch_Size"]),
                                 "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
    out = {"workload": a.workload, "note": a.note, "source": "rocprofv3 --kernel-trace --pmc <set> (one pass per set), tools/pmc_summary.py",
           "kernels": {}}
    for k, e in sorted(per.items()):
        c = {n: sum(v) / len(v) for n, v in e["counters"].items()}
        n_disp = max(len(v) for v in e["counters"].values())
        rec = {"dispatches_per_pass": n_disp, "avg_duration_us_under_pmc": sum(e["dur"]) / max(1, len(e["dur"])) / 1e3, **e["meta"],
               "counters_avg_per_dispatch": c}
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rec["hbm_bytes_per_launch"] = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
            rec["hbm_bytes_note"] = "2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, KB x 1024" + \
                ("" if "FETCH_SIZE" in c and "WRITE_SIZE" in c else " [only one of the two passes present]")
        if c.get("SQ_INSTS_MFMA"):
            rec["valu_per_mfma"] = (c.get("SQ_INSTS_VALU", 0.0) - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
            cyc = 32.0 if "16" in k.split("<")[0] else 64.0
            dur = rec["avg_duration_us_under_pmc"] * 1e-6
            rec["mfma_pipe_util"] = c["SQ_INSTS_MFMA"] * cyc / (dur * 2.4e9 * 1024) if dur > 0 else None
            rec["mfma_cycles_per_instruction_assumed"] = cyc
        if c.get("SQ_WAVE_CYCLES"):
            rec["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            rec["wait_inst_any_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            rec["active_inst_any_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        out["kernels"][k] = rec
    mlps = {k: v for k, v in out["kernels"].items() if k.startswith("mlp_") and "backward" not in k}
    if mlps:
        out["dominant_mlp"] = max(mlps, key=lambda k: mlps[k]["avg_duration_us_under_pmc"] * mlps[k]["dispatches_per_pass"])
    k1 = [k for k in out["kernels"] if k.startswith("scatter_sum_csr_vec4") and k.endswith("false>")] or \
         [k for k in out["kernels"] if k.startswith("scatter_sum_csr")]  # <lanes, PERM = false>: CSR-ordered messages
    if k1:
        out["k1"] = k1[0]
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {a.out}: {len(out['kernels'])} kernels; dominant_mlp={out.get('dominant_mlp')}; k1={out.get('k1')}")


if __name__ == "__main__":
    sys.exit(main())
