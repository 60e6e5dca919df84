"""Minimal PCD (Point Cloud Data v0.7) reader / writer for x,y,z float32 clouds.

Stands in for ``pcl::io::loadPCDFile<pcl::PointXYZ>`` at the reference's map-ingest step
(cpp/.../src/planner/trg_planner.cpp:76-101).  Supports DATA ascii, binary and binary_compressed
(LZF, column-major payload as PCL writes it); extra fields are skipped.
"""
from __future__ import annotations

import struct

import numpy as np

_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16,
       ("U", 4): np.uint32, ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32}


def lzf_decompress(src: bytes, out_len: int) -> bytes:
    """liblzf stream decoder (literal runs and back references)."""
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]
        ip += 1
        if ctrl < 32:  # literal run of ctrl+1 bytes
            ln = ctrl + 1
            out[op:op + ln] = src[ip:ip + ln]
            ip += ln
            op += ln
        else:  # back reference
            ln = ctrl >> 5
            if ln == 7:
                ln += src[ip]
                ip += 1
            ref = op - ((ctrl & 0x1F) << 8) - src[ip] - 1
            ip += 1
            ln += 2
            if ref < 0:
                raise ValueError("corrupt LZF stream")
            for _ in range(ln):  # may overlap
                out[op] = out[ref]
                op += 1
                ref += 1
    if op != out_len:
        raise ValueError("LZF length mismatch")
    return bytes(out)


def lzf_compress_literal(src: bytes) -> bytes:
    """A valid (uncompressed) LZF stream: literal runs only.  Used to write test fixtures."""
    out = bytearray()
    for i in range(0, len(src), 32):
        chunk = src[i:i + 32]
        out.append(len(chunk) - 1)
        out += chunk
    return bytes(out)


def read_pcd(path) -> np.ndarray:
    """Return the cloud as float32 ``(N, 3)`` (x, y, z)."""
    with open(path, "rb") as f:
        raw = f.read()
    header = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        header[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields = header["FIELDS"]
    sizes = [int(v) for v in header["SIZE"]]
    types = header["TYPE"]
    counts = [int(v) for v in header.get("COUNT", ["1"] * len(fields))]
    npts = int(header["POINTS"][0]) if "POINTS" in header else \
        int(header["WIDTH"][0]) * int(header["HEIGHT"][0])
    mode = header["DATA"][0].lower()
    dt = np.dtype([(f, _NP[(t, s)], (c,) if c > 1 else ()) for f, s, t, c in
                   zip(fields, sizes, types, counts)])
    if not all(k in fields for k in ("x", "y", "z")):
        raise ValueError("PCD file lacks x/y/z fields")
    if mode == "ascii":
        txt = raw[pos:].decode("ascii", "replace").split()
        ncol = sum(counts)
        arr = np.array(txt[:npts * ncol], dtype=np.float64).reshape(npts, ncol)
        col = {}
        k = 0
        for f, c in zip(fields, counts):
            col[f] = k
            k += c
        return np.stack([arr[:, col["x"]], arr[:, col["y"]], arr[:, col["z"]]], 1).astype(np.float32)
    if mode == "binary":
        rec = np.frombuffer(raw, dtype=dt, count=npts, offset=pos)
    elif mode == "binary_compressed":
        comp, uncomp = struct.unpack_from("<II", raw, pos)
        blob = lzf_decompress(raw[pos + 8:pos + 8 + comp], uncomp)
        rec = np.empty(npts, dtype=dt)  # payload is stored field by field (SoA)
        off = 0
        for f, s, c in zip(fields, sizes, counts):
            nbytes = npts * s * c
            col = np.frombuffer(blob, dtype=dt[f].base, count=npts * c, offset=off)
            rec[f] = col.reshape(rec[f].shape)
            off += nbytes
    else:
        raise ValueError(f"unsupported PCD DATA mode {mode}")
    return np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float32)


def write_pcd(path, xyz, mode="binary"):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\n"
            f"TYPE F F F\nCOUNT 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\n"
            f"DATA {mode}\n").encode("ascii")
    with open(path, "wb") as f:
        f.write(head)
        if mode == "ascii":
            for p in xyz:
                f.write(("%.9g %.9g %.9g\n" % tuple(p)).encode("ascii"))
        elif mode == "binary":
            f.write(xyz.tobytes())
        elif mode == "binary_compressed":
            blob = np.ascontiguousarray(xyz.T).tobytes()  # x..., y..., z...
            comp = lzf_compress_literal(blob)
            f.write(struct.pack("<II", len(comp), len(blob)))
            f.write(comp)
        else:
            raise ValueError(mode)
