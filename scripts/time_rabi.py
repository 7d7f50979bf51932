"""Development aid: where one evaluation of the 2-level rabi case (57 steps) spends its 0.13 ms -- host wall time per call next to the
propagator time of the HIP events (JQ_DEBUG_TIMING=1 prints the library's own break-down)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import juqbox_jl_amd as jq
p, info = jq.cases.BUILDERS["rabi"]()
pcof = info["pcof0"]
wa = jq.Working_Arrays_HIP(p, pcof.size)
for _ in range(20):
    jq.traceobjgrad(pcof, p, wa)
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    jq.traceobjgrad(pcof, p, wa)
t1 = time.perf_counter()
t = wa.last_timing()
print("rabi: %.1f us per traceobjgrad call (host wall, %d calls); last_timing: %s" % ((t1 - t0) / n * 1e6, n, {k: t[k] for k in ("ms_total", "ms_forward", "ms_backward", "kernel_family", "kernel_variant")}))
t0 = time.perf_counter()
for _ in range(n):
    jq.traceobjgrad(pcof, p, wa, False, False)
t1 = time.perf_counter()
print("rabi: %.1f us per objective-only call" % ((t1 - t0) / n * 1e6))
wa.close()
