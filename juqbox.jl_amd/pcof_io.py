"""Control-vector file formats of the reference, so optimised pulses round-trip with its tooling.

* ``save_pcof`` / ``read_pcof`` mirror ``src/save_pcof.jl:12-28``: a JLD2 file with one dataset ``pcof``
  (Vector{Float64}).  JLD2 files are a strict subset of HDF5 (512-byte text header, version-2 superblock,
  version-2 object headers with Jenkins lookup3 checksums, compact or contiguous data layout); this module
  reads and writes that subset with nothing but ``struct`` + numpy -- h5py is not available in the target
  image.  The writer reproduces ``examples/drives/rabi-pcof-opt-t100.jld2`` byte for byte when given the
  same vector and writer tag (tests/test_pcof_io.py), which pins the structure and the checksums.
* ``read_jld2`` returns every Float64 dataset of the root group (the layout of
  ``test/reference_solutions/*-ref.jld2``: ``obj0``, ``grad0``, ...), skipping the ``_types`` group.
* ``read_dat`` / ``save_dat``: the plain text vectors of ``test/cases/*.dat`` (one number per line).

Host-side utility; nothing here touches the GPU.
"""
import struct

import numpy as np

JLD2_MAGIC = b"HDF5-based Julia Data Format, version "
_UNDEF = 0xFFFFFFFFFFFFFFFF
_F64_DATATYPE = bytes.fromhex("31203f00" "08000000" "0000" "4000" "34" "0b" "00" "34" "ff030000")


def _rot(x, k):
    return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF


def lookup3(data, initval=0):
    """Bob Jenkins' lookup3 ``hashlittle`` as used by HDF5 (H5_checksum_lookup3) for v2 metadata."""
    length = len(data)
    a = b = c = (0xDEADBEEF + length + initval) & 0xFFFFFFFF
    M = 0xFFFFFFFF
    k = 0
    while length > 12:
        a = (a + int.from_bytes(data[k:k + 4], "little")) & M
        b = (b + int.from_bytes(data[k + 4:k + 8], "little")) & M
        c = (c + int.from_bytes(data[k + 8:k + 12], "little")) & M
        a = (a - c) & M; a ^= _rot(c, 4); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 6); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 8); b = (b + a) & M
        a = (a - c) & M; a ^= _rot(c, 16); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 19); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 4); b = (b + a) & M
        length -= 12
        k += 12
    if length == 0:
        return c
    tail = data[k:k + length] + b"\0" * (12 - length)
    a = (a + int.from_bytes(tail[0:4], "little")) & M
    b = (b + int.from_bytes(tail[4:8], "little")) & M
    c = (c + int.from_bytes(tail[8:12], "little")) & M
    c ^= b; c = (c - _rot(b, 14)) & M
    a ^= c; a = (a - _rot(c, 11)) & M
    b ^= a; b = (b - _rot(a, 25)) & M
    c ^= b; c = (c - _rot(b, 16)) & M
    a ^= c; a = (a - _rot(c, 4)) & M
    b ^= a; b = (b - _rot(a, 14)) & M
    c ^= b; c = (c - _rot(b, 24)) & M
    return c


# ------------------------------------------------------------------------------------------------ reader
class _File:
    def __init__(self, data):
        self.d = data
        if not data.startswith(JLD2_MAGIC):
            raise ValueError("not a JLD2 file (missing 'HDF5-based Julia Data Format' header)")
        self.sb = 512
        if data[self.sb:self.sb + 8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError("HDF5 superblock not found at offset 512")
        ver, so, sl = data[self.sb + 8], data[self.sb + 9], data[self.sb + 10]
        if ver not in (2, 3) or so != 8 or sl != 8:
            raise ValueError("unsupported HDF5 superblock (version %d, offsets %d, lengths %d)" % (ver, so, sl))
        self.base, _ext, self.eof, self.root = struct.unpack_from("<QQQQ", data, self.sb + 12)
        if lookup3(data[self.sb:self.sb + 44]) != struct.unpack_from("<I", data, self.sb + 44)[0]:
            raise ValueError("superblock checksum mismatch")

    def messages(self, addr):
        """(type, payload) of every message of the version-2 object header at relative address addr."""
        d = self.d
        p = self.base + addr
        if d[p:p + 4] != b"OHDR" or d[p + 4] != 2:
            raise ValueError("version-2 object header expected at %#x" % p)
        flags = d[p + 5]
        q = p + 6
        if flags & 0x20:
            q += 16            # access/modification/change/birth times
        if flags & 0x10:
            q += 4             # max compact / min dense attributes
        nsz = 1 << (flags & 3)
        size0 = int.from_bytes(d[q:q + nsz], "little")
        q += nsz
        if lookup3(d[p:q + size0]) != struct.unpack_from("<I", d, q + size0)[0]:
            raise ValueError("object header checksum mismatch at %#x" % p)
        blocks = [(q, q + size0)]
        out = []
        creation_order = bool(flags & 0x04)
        while blocks:
            s, e = blocks.pop(0)
            while s + 4 <= e:
                mtype = d[s]
                msize = struct.unpack_from("<H", d, s + 1)[0]
                s += 4 + (2 if creation_order else 0)
                payload = d[s:s + msize]
                s += msize
                if mtype == 0x10:          # continuation: address, length of an OCHK block
                    caddr, clen = struct.unpack_from("<QQ", payload, 0)
                    cp = self.base + caddr
                    if d[cp:cp + 4] != b"OCHK":
                        raise ValueError("object header continuation block expected at %#x" % cp)
                    if lookup3(d[cp:cp + clen - 4]) != struct.unpack_from("<I", d, cp + clen - 4)[0]:
                        raise ValueError("continuation block checksum mismatch at %#x" % cp)
                    blocks.append((cp + 4, cp + clen - 4))
                elif mtype != 0:
                    out.append((mtype, payload))
        return out

    def links(self, addr):
        res = {}
        for mtype, pl in self.messages(addr):
            if mtype != 0x06:
                continue
            ver, fl = pl[0], pl[1]
            if ver != 1:
                raise ValueError("unsupported link message version %d" % ver)
            k = 2
            ltype = 0
            if fl & 0x08:
                ltype = pl[k]; k += 1
            if fl & 0x04:
                k += 8
            if fl & 0x10:
                k += 1
            lsz = 1 << (fl & 3)
            nlen = int.from_bytes(pl[k:k + lsz], "little"); k += lsz
            name = pl[k:k + nlen].decode("utf-8"); k += nlen
            if ltype == 0:
                res[name] = struct.unpack_from("<Q", pl, k)[0]
        return res

    def dataset(self, addr):
        shape = dtype_ok = raw = None
        for mtype, pl in self.messages(addr):
            if mtype == 0x01:              # dataspace
                ver, rank, fl = pl[0], pl[1], pl[2]
                if ver == 2:
                    off = 4
                elif ver == 1:
                    off = 8
                else:
                    raise ValueError("unsupported dataspace version %d" % ver)
                shape = struct.unpack_from("<%dQ" % rank, pl, off) if rank else ()
            elif mtype == 0x03:            # datatype
                dtype_ok = (pl[0] & 0x0F) == 1 and struct.unpack_from("<I", pl, 4)[0] == 8
            elif mtype == 0x08:            # data layout
                ver, cls = pl[0], pl[1]
                if ver not in (3, 4):
                    raise ValueError("unsupported data layout version %d" % ver)
                if cls == 0:
                    n = struct.unpack_from("<H", pl, 2)[0]
                    raw = pl[4:4 + n]
                elif cls == 1:
                    a, n = struct.unpack_from("<QQ", pl, 2)
                    raw = b"" if a == _UNDEF else self.d[self.base + a:self.base + a + n]
                else:
                    raise ValueError("chunked datasets are not supported")
        if not dtype_ok or shape is None or raw is None:
            return None
        arr = np.frombuffer(raw, dtype="<f8").copy()
        if shape == ():
            return float(arr[0])
        # Julia arrays are column-major; HDF5 dimensions are stored slowest-first, JLD2 reverses them
        return arr.reshape(tuple(reversed(shape)), order="F") if len(shape) > 1 else arr


def read_jld2(path):
    """All Float64 datasets (scalars -> float, arrays -> ndarray) of the root group of a JLD2 file."""
    with open(path, "rb") as fh:
        f = _File(fh.read())
    out = {}
    for name, addr in f.links(f.root).items():
        if name.startswith("_"):
            continue
        try:
            v = f.dataset(addr)
        except ValueError:
            v = None
        if v is not None:
            out[name] = v
    return out


def read_pcof(path):
    """``read_pcof(refFileName)`` (src/save_pcof.jl:23-28); '.dat' files are read as text vectors."""
    if str(path).endswith(".dat"):
        return read_dat(path)
    d = read_jld2(path)
    if "pcof" not in d:
        raise KeyError("no dataset 'pcof' in %s" % path)
    return np.asarray(d["pcof"], dtype=float)


# ------------------------------------------------------------------------------------------------ writer
def _ohdr(messages, pad_to=0):
    """Version-2 object header from (type, payload[, message flags]) tuples, NIL-padded to pad_to bytes."""
    body = b"".join(bytes([m[0]]) + struct.pack("<H", len(m[1])) + bytes([m[2] if len(m) > 2 else 0]) + m[1] for m in messages)
    if pad_to > len(body):
        gap = pad_to - len(body) - 4
        body += b"\0" + struct.pack("<H", gap) + b"\0" + b"\0" * gap
    if len(body) < 256:
        head = b"OHDR\x02\x00" + bytes([len(body)])
    else:
        head = b"OHDR\x02\x01" + struct.pack("<H", len(body))
    blk = head + body
    return blk + struct.pack("<I", lookup3(blk))


def save_pcof(path, pcof, writer="Julia 1.5.3 64-bit LE"):
    """``save_pcof(refFileName, pcof)`` (src/save_pcof.jl:12-14): JLD2 file with the dataset ``pcof``.

    Same structure JLD2.jl 0.1.x writes for a Vector{Float64}: dataset object header (fill value, dataspace,
    datatype, compact layout up to 8 KiB else contiguous) followed by the root group (link info, group info,
    one hard link).  ``writer`` only fills the free-text part of the 512-byte header.
    """
    v = np.ascontiguousarray(np.asarray(pcof, dtype="<f8").ravel())
    raw = v.tobytes()
    header = JLD2_MAGIC + b"0.1.1\0 (" + writer.encode("ascii") + b")\0"
    header = header.ljust(512, b"\0")
    dataspace = b"\x02\x01\x00\x01" + struct.pack("<Q", v.size)
    msgs = [(0x05, b"\x03\x09"), (0x01, dataspace)]
    dt = (0x03, _F64_DATATYPE, 1)                  # message flag 1 (constant), as JLD2 writes it
    ds_addr = 48                                   # relative to the base address (512): right after the superblock
    compact = len(raw) < 8192
    tail = b""
    if compact:
        layout = b"\x04\x00" + struct.pack("<H", len(raw)) + raw
        ds = _ohdr(msgs + [dt, (0x08, layout)])
    else:
        probe = _ohdr(msgs + [dt, (0x08, b"\x04\x01" + struct.pack("<QQ", 0, len(raw)))])
        ds = None
    root_msgs = [(0x02, b"\x00\x00" + struct.pack("<QQ", _UNDEF, _UNDEF)), (0x0A, b"\x00\x00"),
                 (0x06, b"\x01\x10\x01\x04pcof" + struct.pack("<Q", ds_addr))]
    root = _ohdr(root_msgs, pad_to=68)
    if compact:
        root_addr = ds_addr + len(ds)
        body = ds + root
    else:
        root_addr = ds_addr + len(probe)
        data_addr = root_addr + len(root)
        ds = _ohdr(msgs + [dt, (0x08, b"\x04\x01" + struct.pack("<QQ", data_addr, len(raw)))])
        body = ds + root
        tail = raw
    eof = 512 + 48 + len(body) + len(tail)
    sb = b"\x89HDF\r\n\x1a\n\x02\x08\x08\x00" + struct.pack("<QQQQ", 512, _UNDEF, eof, root_addr)
    sb += struct.pack("<I", lookup3(sb))
    with open(path, "wb") as fh:
        fh.write(header + sb + body + tail)


def read_dat(path):
    """Text vector, one number per line (test/cases/*.dat, examples/drives/cnot2.dat)."""
    return np.atleast_1d(np.loadtxt(path, dtype=float))


def save_dat(path, pcof):
    with open(path, "w") as fh:
        for x in np.asarray(pcof, dtype=float).ravel():
            fh.write(repr(float(x)) + "\n")
