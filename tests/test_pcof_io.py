"""pcof / reference-solution file formats: JLD2 (HDF5 subset) and .dat text.

The binary fixtures under tests/golden/jld2/ are DATA files of the reference (test/reference_solutions/*.jld2,
examples/drives/*.jld2); the expected numbers in tests/golden/*.json were extracted from the same files with
h5py (tests/golden/make_golden.py), so the pure-Python reader is checked against an independent HDF5
implementation, and the writer against JLD2.jl's own output byte for byte."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(__file__)
J = os.path.join(HERE, "golden", "jld2")


@pytest.fixture(scope="module")
def io():
    import juqbox_jl_amd as jq
    return jq.pcof_io


def test_reader_matches_h5py_extraction(io):
    g = json.load(open(os.path.join(HERE, "golden", "swap02.json")))
    d = io.read_jld2(os.path.join(J, "swap02-ref.jld2"))
    assert sorted(d) == ["grad0", "obj0"]
    assert np.array_equal(np.ravel(d["obj0"]), np.ravel(g["obj0"]))          # bit-exact: same bytes
    assert np.array_equal(np.ravel(d["grad0"]), np.ravel(g["grad0"]))
    g = json.load(open(os.path.join(HERE, "golden", "cnot2-leakieq.json")))
    d = io.read_jld2(os.path.join(J, "cnot2-leakieq-ref.jld2"))
    assert np.array_equal(np.ravel(d["obj0"]), np.ravel(g["obj0"])) and d["obj0"].shape == (2,)
    assert np.array_equal(np.ravel(d["grad0"]), np.ravel(g["grad0"])) and d["grad0"].shape == (160,)


def test_reader_multidimensional_is_column_major(io):
    g = json.load(open(os.path.join(HERE, "golden", "err-mat.json")))
    d = io.read_jld2(os.path.join(J, "err-mat-ref.jld2"))
    em = d["err_mat"]
    assert list(em.shape) == list(g["err_mat_shape_julia"])
    assert np.array_equal(em, np.asarray(g["err_mat"]).reshape(em.shape))


@pytest.mark.parametrize("name,n", [("rabi-pcof-opt-t100.jld2", 6), ("cnot3-pcof-opt.jld2", 270)])
def test_writer_reproduces_jld2_output_byte_for_byte(io, tmp_path, name, n):
    ref = open(os.path.join(J, name), "rb").read()
    pcof = io.read_pcof(os.path.join(J, name))
    assert pcof.shape == (n,)
    tag = ref[ref.index(b"(") + 1:ref.index(b")")].decode()
    out = tmp_path / name
    io.save_pcof(str(out), pcof, writer=tag)
    assert out.read_bytes() == ref


def test_roundtrips_and_errors(io, tmp_path):
    rng = np.random.default_rng(5)
    for n in (1, 7, 1023, 1024, 5000):          # compact (< 8 KiB) and contiguous layouts
        v = rng.standard_normal(n)
        f = tmp_path / ("v%d.jld2" % n)
        io.save_pcof(str(f), v)
        assert np.array_equal(io.read_pcof(str(f)), v)
        f2 = tmp_path / ("v%d.dat" % n)
        io.save_dat(str(f2), v)
        assert np.array_equal(io.read_pcof(str(f2)), v)      # repr() round-trips doubles exactly
    bad = tmp_path / "bad.jld2"
    bad.write_bytes(b"not a jld2 file")
    with pytest.raises(ValueError):
        io.read_pcof(str(bad))
    # a flipped payload byte must trip the object-header checksum
    f = tmp_path / "v7.jld2"
    b = bytearray(f.read_bytes())
    b[0x270] ^= 0x01
    f.write_bytes(bytes(b))
    with pytest.raises((ValueError, KeyError)):
        io.read_pcof(str(f))


def test_lookup3_known_answers(io):
    # Bob Jenkins' lookup3.c driver5: hashlittle("", 0, 0) = 0xdeadbeef; "Four score and seven years ago" -> 0x17770551
    assert io.lookup3(b"", 0) == 0xDEADBEEF
    assert io.lookup3(b"Four score and seven years ago", 0) == 0x17770551
    assert io.lookup3(b"Four score and seven years ago", 1) == 0xCD628161
