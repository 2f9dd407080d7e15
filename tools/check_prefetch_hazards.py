#!/usr/bin/env python3
"""Build-time guard for the inline-asm prefetch loads of csrc/conv_rows.hip.

conv3_rows_kernel issues its next-chunk loads as `asm volatile("global_load_dwordx4 %0, ...")`.  hipcc does not know those
are loads: it believes the destination registers are defined at issue and inserts no s_waitcnt for them; the data really lands
at the hand-written s_waitcnt in front of the LDS write.  Should a future compiler (or an edit that raises register pressure)
copy, spill or otherwise touch such a register between issue and that wait, the kernel silently computes with stale data.

This script compiles conv_rows.hip to gfx950 assembly and proves, per kernel, on the control-flow graph:
  1. no instruction reads or overwrites the destination of a vector-memory load while that load can still be in flight
     (vmcnt is modelled exactly: every vector-memory load, store, atomic and LDS-DMA is one entry in issue order, and
     `s_waitcnt vmcnt(N)` retires all but the N youngest);
  2. the kernel uses no scratch (private segment size 0, no scratch_/buffer_ spill traffic, vgpr_spill_count 0): a spill
     reload counts in vmcnt and would also invalidate the counted waits of the weight LDS-DMA.
Exit status 0 = proven for every conv3_rows kernel; 1 = a violation (printed with its line in the .s).

    python tools/check_prefetch_hazards.py [--keep-asm PATH]
    python tools/check_prefetch_hazards.py --source .../csrc/dense_fused.hip --match chain2_kernel --scratch-only
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

VMEM = re.compile(r"^(global_|buffer_|scratch_|flat_)(load|store|atomic)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LABEL = re.compile(r"^([.\w$]+):")
VMCNT = re.compile(r"vmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse_kernels(asm):
    """-> {kernel name: (instructions [(lineno, mnemonic, operand text)], {label: index}, meta text)}"""
    kernels, cur, name = {}, None, None
    lines = asm.splitlines()
    for no, raw in enumerate(lines, 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        m = LABEL.match(line)
        if m and not line.startswith("."):
            if m.group(1).startswith("_Z") or m.group(1).startswith("conv"):
                name = m.group(1)
                cur = {"ins": [], "labels": {}, "meta": []}
                kernels[name] = cur
            continue
        if cur is None:
            continue
        if m:                                   # local label (.LBBx_y)
            cur["labels"][m.group(1)] = len(cur["ins"])
            if m.group(1).startswith(".Lfunc_end"):
                cur = None
            continue
        if line.startswith("."):
            if cur is not None and line.startswith(".amdhsa_"):
                cur["meta"].append(line)
            continue
        parts = line.split(None, 1)
        cur["ins"].append((no, parts[0], parts[1] if len(parts) > 1 else ""))
    # .amdhsa_ directives follow the function body (after .Lfunc_end): attribute them to the kernel named in .amdhsa_kernel
    meta, active = {}, None
    for raw in lines:
        t = raw.strip()
        if t.startswith(".amdhsa_kernel"):
            active = t.split()[1]
            meta[active] = []
        elif t.startswith(".end_amdhsa_kernel"):
            active = None
        elif active and t.startswith(".amdhsa_"):
            meta[active].append(t)
    for k in kernels:
        kernels[k]["meta"] = meta.get(k, [])
    return kernels


def check_kernel(name, k):
    ins, labels = k["ins"], k["labels"]
    problems = []
    for t in k["meta"]:
        if t.startswith(".amdhsa_private_segment_fixed_size") and int(t.split()[1]) != 0:
            problems.append(f"{name}: uses {t.split()[1]} bytes of scratch per lane (register spills)")
    seen = set()
    work = [(0, ())]                             # (instruction index, queue of outstanding vmem ops: tuple of frozenset(dest regs))
    while work:
        pc, q = work.pop()
        while pc < len(ins):
            key = (pc, q)
            if key in seen:
                break
            seen.add(key)
            no, op, args = ins[pc]
            if op.startswith("scratch_"):
                problems.append(f"{name}: line {no}: scratch access `{op} {args}`")
            pending = set().union(*(d for d, _ in q)) if q else set()
            pending_other = set().union(*(d for d, io in q if not io)) if q else set()     # destinations of in-flight loads that are NOT global_/buffer_ loads
            if op == "s_waitcnt":
                m = VMCNT.search(args)
                if m:
                    n = int(m.group(1))
                    q = q[len(q) - n:] if n < len(q) else q
                    if n == 0:
                        q = ()
            elif VMEM.match(op):
                is_lds_dma = "_lds_" in op or args.rstrip().endswith(" lds")
                is_load = "_load" in op and not is_lds_dma
                first, _, rest = args.partition(",")
                dest = regs_of(first) if is_load else set()
                used = regs_of(rest if is_load else args)
                # A load whose DESTINATION overlaps that of an earlier load still in flight is not a hazard on gfx9-family hardware: vector-memory
                # reads return in issue order (that order is what vmcnt counts), so the younger load's data lands last.  hipcc produces the pattern
                # when the earlier value is dead on some path (round 3: the epilogue's skip loads of a row whose stores are predicated off).  What
                # must not happen is an instruction READING such a register, or using it as an address, before the wait.
                # That in-order argument holds among global_* / buffer_* loads only (one return queue, the one vmcnt counts): a flat_* load may
                # be served by the LDS path and return out of order with respect to them, so for any other pairing the destination overlap is
                # still reported.
                in_order = op.startswith(("global_load", "buffer_load"))
                bad = (used | (set() if in_order else dest)) & pending
                if in_order and dest & pending_other:
                    bad |= dest & pending_other
                if bad:
                    problems.append(f"{name}: line {no}: `{op} {args}` touches v{sorted(bad)} while a load into it may be in flight")
                q = (q + ((frozenset(dest), in_order),))[-64:]
            else:
                bad = regs_of(args) & pending
                if bad:
                    problems.append(f"{name}: line {no}: `{op} {args}` touches v{sorted(bad)} while a load into it may be in flight")
            if op == "s_endpgm":
                break
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = args.strip().split()[-1]
                if tgt in labels:
                    work.append((labels[tgt], q))
                if op == "s_branch":
                    break
            pc += 1
            if len(problems) > 20:
                return problems
    return problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep-asm", default=None)
    ap.add_argument("--source", default=os.path.join(CSRC, "conv_rows.hip"))
    ap.add_argument("--match", default="conv3_rows_kernel")
    ap.add_argument("--scratch-only", action="store_true",
                    help="only prove that the matching kernels use no scratch (csrc/dense_fused.hip: a spill reload in a loader wave would "
                         "count in the vmcnt its counted waits rely on)")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        out = args.keep_asm or os.path.join(td, "conv_rows.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", args.source, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    kernels = {n: k for n, k in parse_kernels(asm).items() if args.match in n}
    if not kernels:
        print(f"no kernel matching '{args.match}' found in the assembly", file=sys.stderr)
        return 1
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s*(\d+)", asm)]
    problems = []
    if any(spills):
        problems.append(f"vgpr_spill_count is non-zero somewhere in {os.path.basename(args.source)}: {spills}")
    for n, k in kernels.items():
        if args.scratch_only:
            problems += [f"{n}: uses {t.split()[1]} bytes of scratch per lane (register spills)" for t in k["meta"]
                         if t.startswith(".amdhsa_private_segment_fixed_size") and int(t.split()[1]) != 0]
            problems += [f"{n}: line {no}: scratch access `{op} {a}`" for no, op, a in k["ins"] if op.startswith("scratch_")][:5]
        else:
            problems += check_kernel(n, k)
    for p in problems:
        print("HAZARD:", p)
    print(f"{len(kernels)} kernels checked, {sum(len(k['ins']) for k in kernels.values())} instructions, {len(problems)} problems")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
