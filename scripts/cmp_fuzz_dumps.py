"""Compare two FUZZ_DUMP files of scripts/fuzz_gpu.py (the same draws evaluated by two builds of the library): bit-wise, and for the
draws that differ the largest relative difference.  python scripts/cmp_fuzz_dumps.py a.json b.json"""
import json, sys
import numpy as np
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
assert a.keys() == b.keys(), "different draws"
same = diff = skipped = 0
worst, worst_case = 0.0, None
fams = {}
for k in sorted(a, key=int):
    ra, rb = a[k], b[k]
    if len(ra) == 1 or len(rb) == 1:
        assert ra == rb, (k, ra, rb)
        skipped += 1
        continue
    fams[ra[4]] = fams.get(ra[4], 0) + 1
    if ra == rb:
        same += 1
        continue
    diff += 1
    va = np.array([float(ra[0]), float(ra[1])] + [float(x) for x in ra[2]] + [float(x) for x in ra[3]])
    vb = np.array([float(rb[0]), float(rb[1])] + [float(x) for x in rb[2]] + [float(x) for x in rb[3]])
    e = max(abs(va[0] - vb[0]) / max(abs(vb[0]), 1e-300), abs(va[1] - vb[1]) / max(abs(vb[1]), 1e-3), np.linalg.norm(va[2:] - vb[2:]) / max(np.linalg.norm(vb[2:]), 1e-300))
    if e > worst:
        worst, worst_case = e, (k, ra[4], rb[4])
    if e > 1e-9:
        print("draw %s (families %d / %d): relative difference %.2e   <<<<<< MISMATCH" % (k, ra[4], rb[4], e))
print("%d draws: %d bit-identical, %d differ (largest relative difference %.2e, draw %s), %d without kernels; per kernel family: %s" % (
    len(a), same, diff, worst, worst_case, skipped, dict(sorted(fams.items()))))
