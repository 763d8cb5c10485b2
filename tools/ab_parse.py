#!/usr/bin/env python3
"""Prints the numeric fields of tools/ab_libs.sh output lines side by side: python tools/ab_parse.py file key [key ...]"""
import json, sys
keys = sys.argv[2:]
for l in open(sys.argv[1]):
    if " | {" not in l:
        continue
    lib, js = l.split(" | ", 1)
    try:
        d = json.loads(js)
    except Exception:
        continue
    name = lib.split("/")[-2] if "/" in lib else lib
    print(name.ljust(14), " ".join(f"{k}={d.get(k) if not isinstance(d.get(k), float) else round(d.get(k), 4)}" for k in keys), str(d.get("digest", ""))[:12])
