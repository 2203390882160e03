#!/usr/bin/env python3
"""Static instruction census of one kernel from `hipcc -S` output, per basic block:
    python tools/isa_stats.py file.s <mangled-name-substring>
classes: mfma, valu (incl. v_permlane / v_readlane / v_writelane), vmem_ld, vmem_st, lds, salu, smem, wait, branch.
Prints one row per basic block (label, counts) and the totals; loops show up as backward branches (`-> label` column)."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")):
        return "vmem_ld"
    if op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "buffer_atomic", "global_atomic")):
        return "vmem_st"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    cols = ["mfma", "valu", "swap", "vmem_ld", "vmem_st", "lds", "salu", "smem", "wait", "branch"]
    blocks, cur, order = {}, "entry", ["entry"]
    blocks[cur] = {c: 0 for c in cols}
    blocks[cur]["to"] = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith((".Lfunc_end", ".section", ".amdhsa_kernel")):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            cur = m.group(1)
            order.append(cur)
            blocks[cur] = {c: 0 for c in cols}
            blocks[cur]["to"] = []
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        if c == "other":
            continue
        if op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")):
            blocks[cur]["swap"] += 1
        blocks[cur][c] += 1
        if c == "branch":
            m = re.search(r"(\.LBB[0-9_]+)", t)
            if m:
                blocks[cur]["to"].append(m.group(1))
            # a conditional branch ends the basic block: what follows falls through into an unnamed block
            nxt = cur.split("+")[0] + "+%d" % (int(cur.split("+")[1]) + 1 if "+" in cur else 1)
            cur = nxt
            order.append(cur)
            blocks[cur] = {c2: 0 for c2 in cols}
            blocks[cur]["to"] = []
    tot = {c: 0 for c in cols}
    print("%-14s" % "block" + "".join("%8s" % c for c in cols) + "  branches")
    pos = {b: i for i, b in enumerate(order)}
    for b in order:
        r = blocks[b]
        if sum(r[c] for c in cols) == 0:
            continue
        for c in cols:
            tot[c] += r[c]
        to = " ".join(("^" if pos.get(x, 1 << 30) <= pos[b.split("+")[0]] else "v") + x for x in r["to"])
        print("%-14s" % b + "".join("%8d" % r[c] for c in cols) + "  " + to)
    print("%-14s" % "TOTAL" + "".join("%8d" % tot[c] for c in cols))


if __name__ == "__main__":
    main()
