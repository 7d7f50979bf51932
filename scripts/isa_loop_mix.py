#!/usr/bin/env python3
"""Static instruction mix of the time loop (the longest backward branch) of a kernel in a gfx950 .s file.
usage: isa_loop_mix.py <file.s> [kernel name prefix, default _Z10k_backward]"""
import collections
import re
import sys


def mix(path, prefix="_Z10k_backward"):
    lines = open(path).read().splitlines()
    start = [i for i, l in enumerate(lines) if l.startswith(prefix)][0]
    end = [i for i, l in enumerate(lines) if "s_endpgm" in l and i > start][0]
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        if re.match(r"\s+s_c?branch", l):
            t = l.split()[-1]
            if t in labels and labels[t] < i and i - labels[t] > best[0]:
                best = (i - labels[t], labels[t], i)
    loop = body[best[1]:best[2] + 1]
    cnt = collections.Counter()
    for l in loop:
        l = l.strip()
        if not l or l.startswith(";") or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        cls = ("mfma" if "mfma" in op else "scratch" if op.startswith("scratch") else "dpp" if "dpp" in l else "branch" if "branch" in op
               else "barrier" if "barrier" in op else "waitcnt" if "waitcnt" in op else "nop" if op == "s_nop" else "lds" if op.startswith("ds_")
               else "vmem" if op.startswith(("global", "flat", "buffer")) else "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "other")
        cnt[cls] += 1
        cnt["total"] += 1
    return dict(cnt)


if __name__ == "__main__":
    print(mix(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "_Z10k_backward"))
