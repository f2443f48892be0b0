#!/usr/bin/env python3
"""Count instructions per basic block of a kernel in the compiler's assembly listing.

    make -C cuda_path_tracer_amd asm          # writes cuda_path_tracer_amd/kernels.s
    python tools/count_fp64_isa.py [kernels.s] [--kernel SUBSTRING ...] [--min-f64 N]

For every kernel whose mangled name contains one of the substrings, the body is cut into basic blocks at its labels
and branches and each block's instructions are counted by class (a block named LABEL+k is the k-th fall-through piece
after LABEL): fp64 VALU (v_*_f64, plus v_cmp/v_cndmask on 64-bit operands are
listed under their own classes), other VALU, SALU, LDS, vector memory.  The blocks with many fp64 instructions are
the kernel's arithmetic: in k_extend_pw the box code (both child boxes of one node; the loop body is unrolled
CPT_PW_BOX_REPEAT times, so it appears that many times) and the triangle code.  bench.py's fp64-issue roofline
prices a node expansion and a triangle test with these counts.
"""
import argparse
import re
import sys


def classify(op):
    if op.startswith("v_") and (op.endswith("_f64") or "_f64_" in op):
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def kernels(text):
    """name -> list of body lines (between 'name:' and its .Lfunc_end)."""
    out = {}
    name = None
    body = []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                out[name] = body
                name = None
            else:
                body.append(line)
    return out


def blocks(body):
    cur, acc, out = "entry", [], []
    for line in body:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            out.append((cur, acc))
            cur, acc = m.group(1), []
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        acc.append(op)
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):  # a branch ends the block; what follows is its fall-through
            out.append((cur, acc))
            n_fall = sum(1 for (l, _) in out if l.split("+")[0] == cur.split("+")[0])
            cur, acc = "%s+%d" % (cur.split("+")[0], n_fall), []
    out.append((cur, acc))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing", nargs="?", default="cuda_path_tracer_amd/kernels.s")
    ap.add_argument("--kernel", action="append", default=[])
    ap.add_argument("--min-f64", type=int, default=20)
    args = ap.parse_args()
    want = args.kernel or ["k_extend_pwILb0ELb0ELb1EE", "k_extend_pwILb0ELb0ELb0EE"]
    ks = kernels(open(args.listing).read())
    for name, body in ks.items():
        if not any(w in name for w in want):
            continue
        print("kernel %s" % name)
        print("  %-12s %6s %6s %6s %6s %6s %6s   fp64 opcodes" % ("block", "f64", "valu", "salu", "lds", "vmem", "all"))
        tot = {}
        for label, ops in blocks(body):
            c = {}
            for op in ops:
                k = classify(op)
                c[k] = c.get(k, 0) + 1
                tot[k] = tot.get(k, 0) + 1
            if c.get("f64", 0) >= args.min_f64:
                kinds = {}
                for op in ops:
                    if classify(op) == "f64":
                        kinds[op] = kinds.get(op, 0) + 1
                desc = " ".join("%s:%d" % kv for kv in sorted(kinds.items(), key=lambda kv: -kv[1]))
                print("  %-12s %6d %6d %6d %6d %6d %6d   %s" % (label, c.get("f64", 0), c.get("valu", 0), c.get("salu", 0),
                                                                 c.get("lds", 0), c.get("vmem", 0), len(ops), desc))
        print("  %-12s %6d %6d %6d %6d %6d %6d" % ("whole kernel", tot.get("f64", 0), tot.get("valu", 0), tot.get("salu", 0),
                                                   tot.get("lds", 0), tot.get("vmem", 0), sum(tot.values())))
    return 0


if __name__ == "__main__":
    sys.exit(main())
