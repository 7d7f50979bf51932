"""Kernel experiments, latency path: one cnot3 evaluation with every library variant of scripts/exp_variants.sh."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
script = os.environ.get("EXP_SCRIPT", "time_latency.py")
tags = sys.argv[1:] or sorted(os.path.basename(p)[6:-3] for p in glob.glob(os.path.join(ROOT, "juqbox.jl_amd/exp/libjq_*.so")))
for tag in tags:
    env = dict(os.environ, JQ_LIB=os.path.join(ROOT, "juqbox.jl_amd/exp/libjq_%s.so" % tag))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], capture_output=True, text=True, env=env)
    print(tag, "\n   ".join(r.stdout.splitlines()[-int(os.environ.get("EXP_LINES", "1")):] or [r.stderr[-300:]]), flush=True)
