"""GPU (-m gpu): the fence around the VGPR register form.  Most MFMA kernel objects are built with an internal LLVM switch
(-mllvm -amdgpu-mfma-vgpr-form=1, csrc/Makefile: worth 10 ... 80 % on the slab kernels, profiles/r06_register_forms.txt) that miscompiled
one object in round 5.  build() also builds the SAME sources with every object in hipcc's default form (libjuqbox_hip_df.so); here a
fixed-seed slice of random problems -- general draws and draws forced onto the slab kernels, the families that ship in VGPR form -- goes
through both libraries and every draw must agree bit for bit.  `make -C juqbox.jl_amd/csrc check-forms` runs 3 x 2 000 draws of the same."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_default_form_library_is_built():
    """the fence is part of the product: a snapshot without the default-form twin fails instead of skipping the check"""
    import check_forms
    assert os.path.exists(check_forms.DF), "juqbox.jl_amd/libjuqbox_hip_df.so missing: __graft_entry__.build() / make check-forms-lib"


def test_vgpr_form_objects_equal_their_default_form_twins_bit_for_bit():
    import check_forms
    total, bad = check_forms.run(350, quick=True, verbose=False)
    assert total == 700 and not bad, bad[:10]


def test_both_libraries_are_builds_of_the_same_sources():
    """jq_version() carries the hash of the sources AND of the flags: the two builds differ in it, the manifest of the default-form
    build has no object in VGPR form, the shipped one has (else the fence guards nothing)"""
    import json
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import juqbox_jl_amd as jq; from juqbox_jl_amd import _lib; L = _lib.load();"
            "print(L.jq_version().decode())" % ROOT)
    import check_forms
    vers = []
    for lib in (check_forms.MAIN, check_forms.DF):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, JQ_LIB=lib), capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-1000:]
        vers.append(r.stdout.strip().splitlines()[-1])
    assert all(v.startswith("gfx950 juqbox_hip") for v in vers), vers
    src = [v.split("src:")[1].split()[0] for v in vers]
    code = [v.split("code:")[1].split()[0] for v in vers]
    assert src[0] != src[1], "the twin was built with the same flags: the fence compares a library with itself"
    assert code[0] == code[1], "the default-form twin was built from OTHER sources (rebuild: make -C juqbox.jl_amd/csrc check-forms-lib): %s" % vers
    man = json.load(open(os.path.join(ROOT, "juqbox.jl_amd", "csrc", "build", "manifest.json"))) if os.path.exists(
        os.path.join(ROOT, "juqbox.jl_amd", "csrc", "build", "manifest.json")) else None
    if man:      # (the build directory does not travel to the GPU box; the CPU suite checks the manifest itself)
        assert any(o.get("vgpr_form") for o in man["objects"].values())
