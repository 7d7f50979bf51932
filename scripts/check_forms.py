"""The fence around the VGPR register form (csrc/Makefile `make check-forms`, tests/test_gpu_forms.py): fixed-seed random problems --
scripts/fuzz_gpu.py in its DIFFERENTIAL mode (no oracle; every draw's objective, leak and gradients are dumped in exact decimal form) --
through the shipped library and through juqbox.jl_amd/libjuqbox_hip_df.so, the default-register-form build of the SAME sources
(`make -C juqbox.jl_amd/csrc check-forms-lib`).  The two must agree BIT FOR BIT in every draw: the kernels perform the same operations in
the same order, only the register allocation differs -- round 5's miscompiled object (w_6_5 in VGPR form) differed in the 4th digit.
usage: check_forms.py [draws per focus = 2000] [--quick]      (focus modes: general, slab, wfull; --quick: general + slab only)
Exit code 1 on the first difference, 2 when a library is missing."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (FORMS_A / FORMS_B: compare two OTHER builds with the same machinery -- A/B checks of kernel changes that must not change a bit)
MAIN = os.environ.get("FORMS_A") or os.path.join(ROOT, "juqbox.jl_amd", "libjuqbox_hip.so")
DF = os.environ.get("FORMS_B") or os.path.join(ROOT, "juqbox.jl_amd", "libjuqbox_hip_df.so")


def dump(lib, focus, n, seed, path):
    env = dict(os.environ, JQ_LIB=lib, FUZZ_DUMP=path)
    env.pop("FUZZ_FOCUS", None)
    if focus:
        env["FUZZ_FOCUS"] = focus
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_gpu.py"), str(n), str(seed)], env=env, cwd=ROOT, capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(path):
        raise RuntimeError("fuzz_gpu.py failed with %s:\n%s" % (lib, (r.stdout + r.stderr)[-2000:]))
    return json.load(open(path))


def run(n=2000, quick=False, verbose=True):
    """returns (draws compared, list of differing draws)"""
    for lib in (MAIN, DF):
        if not os.path.exists(lib):
            raise FileNotFoundError(lib)
    total, bad = 0, []
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        for focus, seed in ((None, 6101), ("slab", 6102)) + (() if quick else (("wfull", 6103),)):
            a = dump(MAIN, focus, n, seed, os.path.join(tmp, "a.json"))
            b = dump(DF, focus, n, seed, os.path.join(tmp, "b.json"))
            assert a.keys() == b.keys() and len(a) == n, (len(a), len(b))
            fam = {}
            for k in a:
                total += 1
                if a[k] != b[k]:
                    bad.append((focus or "general", int(k)))
                if len(a[k]) == 5:
                    fam[a[k][4]] = fam.get(a[k][4], 0) + 1
            if verbose:
                print("focus %-8s seed %d: %d draws, kernel families %s, %d differ  (%.0f s)" % (focus or "general", seed, n,
                      " ".join("%d:%d" % kv for kv in sorted(fam.items())), sum(1 for f, _ in bad if f == (focus or "general")), time.time() - t0), flush=True)
    return total, bad


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    try:
        total, bad = run(int(args[0]) if args else 2000, quick="--quick" in sys.argv)
    except FileNotFoundError as e:
        print("missing library: %s (make -C juqbox.jl_amd/csrc check-forms-lib)" % e)
        sys.exit(2)
    print("%d draws through both register forms: %s" % (total, "bit-identical in every one" if not bad else "%d DIFFER: %s" % (len(bad), bad[:10])))
    sys.exit(1 if bad else 0)
