#!/usr/bin/env python3
"""Loop map of one kernel's gfx950 assembly (hipcc -S): every backward branch is a loop [label, branch]; prints
per loop its size, FP64 / scratch / LDS / global instruction counts, so that spills inside hot loops stand out.
usage: asm_loops.py <kernel.s> [min_lines]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
minl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i, t))
loops.sort(key=lambda x: (x[0], -x[1]))


def isinstr(l):
    l = l.strip()
    return bool(l) and not l.startswith((";", ".", "/")) and not l.endswith(":")


print("%-14s %7s %7s %6s %6s %6s %6s %6s %6s depth" % ("loop", "start", "end", "instr", "f64", "sload", "sstore", "lds", "glob"))
for a, b, t in loops:
    body = [l for l in lines[a:b + 1] if isinstr(l)]
    if len(body) < minl:
        continue
    depth = sum(1 for (c, d, _) in loops if c <= a and d >= b) - 1
    print("%-14s %7d %7d %6d %6d %6d %6d %6d %6d %d" % (
        t, a, b, len(body), sum("f64" in l for l in body), sum("scratch_load" in l for l in body),
        sum("scratch_store" in l for l in body), sum(l.strip().startswith("ds_") for l in body),
        sum(l.strip().startswith(("global_", "buffer_")) for l in body), depth))
