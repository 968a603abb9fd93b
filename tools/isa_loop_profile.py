#!/usr/bin/env python3
"""Developer view of one kernel's main loop in the built code object: instruction mix and every s_waitcnt / s_barrier
with the number of matrix instructions issued since the previous one.

    python tools/isa_loop_profile.py mlp_backward.o 'mlp_backward_fused_kernel<2>' [--waits] [--dump]
"""
import argparse
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        return "lds_read"
    if op.startswith("ds_"):
        return "lds_write"
    if re.match(r"(buffer|global|flat)_load", op):
        return "vmem_load"
    if re.match(r"(buffer|global|flat)_(store|atomic)", op):
        return "vmem_store"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("v_accvgpr"):
        return "accvgpr_mov"
    if op.startswith("v_"):
        return "valu"
    if op in ("s_waitcnt", "s_barrier", "s_nop"):
        return op
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    return "salu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("kernel")
    ap.add_argument("--waits", action="store_true")
    ap.add_argument("--dump", action="store_true")
    a = ap.parse_args()
    obj = a.obj if os.path.exists(a.obj) else os.path.join(ROOT, "build", "csrc", a.obj)
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co)], stdout=subprocess.PIPE, text=True).stdout
    filt = shutil.which("c++filt")
    name, ins, found = None, [], None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            if found:
                break
            d = subprocess.run([filt, m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip() if filt else m.group(1)
            d = d.replace("(anonymous namespace)::", "").replace("void ", "")
            if d.startswith(a.kernel):
                found = d
            continue
        if found:
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):[^<]*(<.*>)?", line)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4) or ""))
    if not found:
        sys.exit("kernel not found")
    base = ins[0][0]
    loops = []
    for addr, op, args, tgt in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"\+0x([0-9a-f]+)>", tgt)
            t = base + int(m.group(1), 16) if m else (base if tgt else None)
            if t is not None and t <= addr:
                loops.append((t, addr))
    mf = [x[0] for x in ins if x[1].startswith("v_mfma")]
    cover = [l for l in loops if mf and sum(1 for m_ in mf if l[0] <= m_ <= l[1]) >= 0.9 * len(mf)]
    # the persistent tile loop: the SMALLEST backward-branch range holding >= 90 % of the kernel's matrix instructions
    main_loop = min(cover, key=lambda r: r[1] - r[0]) if cover else max(loops, key=lambda r: r[1] - r[0])
    body = [x for x in ins if main_loop[0] <= x[0] <= main_loop[1]]
    hist = collections.Counter(classify(op) for _, op, _, _ in body)
    nops = sum(int(args or 0) + 1 for _, op, args, _ in body if op == "s_nop")
    print(f"{found.split('(')[0]}: {len(ins)} instructions, main loop {len(body)} (0x{main_loop[0]:x}..0x{main_loop[1]:x})")
    print("  mix:", dict(sorted(hist.items(), key=lambda kv: -kv[1])), "| s_nop wait states:", nops)
    inner = [l for l in loops if main_loop[0] <= l[0] and l[1] <= main_loop[1] and l != main_loop]
    print(f"  inner loops: {len(inner)}")
    if a.waits:
        since = 0
        for addr, op, args, _ in body:
            if op.startswith("v_mfma"):
                since += 1
            if op in ("s_waitcnt", "s_barrier"):
                if "vmcnt" in args or op == "s_barrier":
                    print(f"    0x{addr:x} {op} {args}   (+{since} mfma)")
                    since = 0
    if a.dump:
        for addr, op, args, _ in body:
            print(f"{addr:x}\t{op} {args}")


if __name__ == "__main__":
    main()
