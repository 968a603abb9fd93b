#!/usr/bin/env python3
"""ISA guard for the hand-scheduled kernels (run by __graft_entry__.build(); exits non-zero on a violation).

The FAST kernels issue their output stores from inline asm so that hipcc's `vmcnt` bookkeeping does not see them
(DESIGN.md, K4): that is what lets prefetched loads stay in flight, and it is also what takes three guarantees away
from the compiler.  This script re-establishes them on the built code objects (gfx950 disassembly of every
build/csrc/*.o):

  (a) drained waits: `s_waitcnt vmcnt(0)` inside a kernel's MAIN loop (the smallest backward-branch range that holds
      90 % of its matrix instructions = the persistent tile loop) may not exceed the recorded budget - a new one means some change made the compiler give
      up counted waits again and every prefetch of that tile is drained at that point;
  (b) SGPR hazard: a VMEM instruction (buffer_/global_/flat_/scratch_ load or store) that reads an SGPR written by a
      VALU instruction (`v_readlane_b32` / `v_readfirstlane_b32`: SGPR-spill restores, wave-uniform values) needs 5
      wait states in between (gfx9 ISA: "VALU writes SGPR -> VMEM reads that SGPR"); the hazard recognizer does not
      look inside inline asm, so this is checked for EVERY such pair, counting each instruction as one wait state
      and `s_nop N` as N + 1;
  (c) scratch: `.private_segment_fixed_size` (and the spill counts behind it) per kernel may not exceed the recorded
      budget.

Budgets live in tools/isa_budget.json (`python tools/check_isa.py --update` rewrites them from the current build;
review the diff before committing it).
"""
import argparse
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
BUDGET = os.path.join(ROOT, "tools", "isa_budget.json")
VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")
HAZARD_WAIT_STATES = int(os.environ.get("GNC_ISA_HAZARD_STATES", "5"))
SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32")


def run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True, **kw).stdout


def demangle(names):
    filt = os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(filt):
        filt = shutil.which("c++filt") or filt
    if not os.path.exists(filt) or not names:
        return {n: n for n in names}
    out = run([filt] + list(names)).splitlines()
    return {n: d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for n, d in zip(names, out)}


def sgprs_of(text):
    """SGPR numbers named in an operand string."""
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bs(\d+)\b", text))
    return regs


def analyse_object(obj, tmp):
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp)
    cos = [f for f in glob.glob(local + ".*") if "amdgcn" in f]
    kernels = {}
    for co in cos:
        notes = run([os.path.join(LLVM, "llvm-readelf"), "--notes", co])
        meta, cur = {}, {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip()
            if key == "agpr_count" and cur.get("name"):
                meta[cur["name"]] = cur
                cur = {}
            if key in ("name", "private_segment_fixed_size", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count",
                       "sgpr_spill_count", "group_segment_fixed_size"):
                cur[key] = val if key == "name" else int(val)
        if cur.get("name"):
            meta[cur["name"]] = cur
        dis = run([os.path.join(LLVM, "llvm-objdump"), "-d", co])
        name, insts = None, []
        blocks = {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
            if m:
                if name:
                    blocks[name] = insts
                name, insts = m.group(1), []
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):[^<]*(<.*>)?", line)
            if m and name:
                insts.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4) or ""))
        if name:
            blocks[name] = insts
        for kname, ins in blocks.items():
            if kname not in meta:
                continue
            base = ins[0][0] if ins else 0
            # loops = backward branches
            loops = []
            for addr, op, args, tgt_txt in ins:
                if op.startswith("s_cbranch") or op == "s_branch":
                    m = re.search(r"\+0x([0-9a-f]+)>", tgt_txt)
                    tgt = base + int(m.group(1), 16) if m else (base if tgt_txt else None)
                    if tgt is not None and tgt <= addr:
                        loops.append((tgt, addr))
            mf = [x[0] for x in ins if x[1].startswith("v_mfma")]
            cover = [l for l in loops if mf and sum(1 for m_ in mf if l[0] <= m_ <= l[1]) >= 0.9 * len(mf)]
            # the persistent tile loop: the SMALLEST backward-branch range holding >= 90 % of the kernel's matrix
            # instructions (kernels without MFMAs: the largest range)
            main = (min(cover, key=lambda r: r[1] - r[0]) if cover else max(loops, key=lambda r: r[1] - r[0])) if loops else None
            vm0 = sum(1 for addr, op, args, _ in ins if main and main[0] <= addr <= main[1] and op == "s_waitcnt"
                      and re.search(r"vmcnt\(0\)", args))
            # hazard (b)
            hazards = []
            last_write = {}  # sgpr -> wait states since a VALU wrote it
            for addr, op, args, _ in ins:
                if VMEM.match(op):
                    ops = args.split(",")
                    read = sgprs_of(",".join(ops[1:]) if "store" in op or "atomic" in op else ",".join(ops[1:]))
                    for r in read:
                        if r in last_write and last_write[r] < HAZARD_WAIT_STATES:
                            hazards.append(f"{op} at 0x{addr:x} reads s{r} {last_write[r]} wait states after a VALU wrote it")
                step = 1
                if op == "s_nop":
                    step = int(args.strip() or 0) + 1
                for r in list(last_write):
                    last_write[r] += step
                    if last_write[r] > 16:
                        del last_write[r]
                if op in SGPR_WRITERS:
                    dst = args.split(",")[0]
                    for r in sgprs_of(dst):
                        last_write[r] = 0
                if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
                    pass  # linear scan: conservative across fall-through, branches do not shorten the distance
            m_ = meta[kname]
            kernels[kname] = {"scratch_bytes": m_.get("private_segment_fixed_size", 0), "vgpr": m_.get("vgpr_count", 0),
                              "agpr": m_.get("agpr_count", 0), "sgpr_spills": m_.get("sgpr_spill_count", 0),
                              "vgpr_spills": m_.get("vgpr_spill_count", 0), "vmcnt0_in_main_loop": vm0, "hazards": hazards,
                              "instructions": len(ins)}
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--update", action="store_true", help="rewrite tools/isa_budget.json from the current build")
    ap.add_argument("--objects", default=os.path.join(ROOT, "build", "csrc", "*.o"))
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    objs = sorted(glob.glob(a.objects))
    mk = os.path.join(ROOT, "graphnet_classifier_amd", "csrc", "Makefile")
    m = re.search(r"^SRCS\s*=\s*(.+)$", open(mk).read(), re.M) if os.path.exists(mk) else None
    if m:  # only the objects the shared library links: probe / variant builds left in build/csrc are not shipped
        shipped = {s[:-4] + ".o" for s in m.group(1).split() if s.endswith(".hip")}
        objs = [o for o in objs if os.path.basename(o) in shipped]
    if not objs:
        print("check_isa: no objects under build/csrc (run make first)")
        return 1
    allk = {}
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            allk.update(analyse_object(o, tmp))
    names = demangle(list(allk))
    table = {names[k]: v for k, v in allk.items()}
    if a.update:
        budget = {k: {"scratch_bytes": v["scratch_bytes"], "vmcnt0_in_main_loop": v["vmcnt0_in_main_loop"],
                      "vgpr": v["vgpr"], "agpr": v["agpr"]} for k, v in sorted(table.items())}
        with open(BUDGET, "w") as f:
            json.dump(budget, f, indent=1, sort_keys=True)
        print(f"check_isa: wrote {BUDGET} ({len(budget)} kernels)")
    budget = json.load(open(BUDGET)) if os.path.exists(BUDGET) else {}
    bad = []
    for k, v in sorted(table.items()):
        b = budget.get(k)
        if v["hazards"]:
            bad += [f"{k}: {h}" for h in v["hazards"]]
        if b is None:
            if v["scratch_bytes"] or v["vmcnt0_in_main_loop"]:
                bad.append(f"{k}: not in tools/isa_budget.json but has scratch {v['scratch_bytes']} B / "
                           f"{v['vmcnt0_in_main_loop']} drained waits in its main loop (run --update and review)")
            continue
        if v["scratch_bytes"] > b["scratch_bytes"]:
            bad.append(f"{k}: scratch {v['scratch_bytes']} B > budget {b['scratch_bytes']} B")
        if v["vmcnt0_in_main_loop"] > b["vmcnt0_in_main_loop"]:
            bad.append(f"{k}: {v['vmcnt0_in_main_loop']} `s_waitcnt vmcnt(0)` in the main loop > budget {b['vmcnt0_in_main_loop']}")
        if a.verbose:
            print(f"{k}: vgpr {v['vgpr']} agpr {v['agpr']} scratch {v['scratch_bytes']} B, vmcnt(0) in main loop {v['vmcnt0_in_main_loop']}")
    if bad:
        print("check_isa: FAILED")
        for b_ in bad:
            print("  " + b_)
        return 1
    tot_scr = sum(1 for v in table.values() if v["scratch_bytes"])
    print(f"check_isa: ok ({len(table)} kernels, {tot_scr} with scratch within budget, no SGPR-write -> VMEM hazards, "
          f"no new drained waits in the main loops)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
