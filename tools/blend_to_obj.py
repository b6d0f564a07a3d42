#!/usr/bin/env python3
"""Reconstructs the OBJ meshes the reference expects but does not ship.

The reference loads `bunny.obj`, `sword_fixed.obj`, ... (main.cpp:409-412) but its .gitignore drops
every *.obj, so only the Blender sources survive.  This tool reads those uncompressed
`BLENDER-v277` files with a small DNA-driven parser (block table + the DNA1 struct catalogue), and
writes what Blender's OBJ exporter would: object matrix applied, axes mapped (x, y, z) -> (x, z, -y),
faces triangulated as fans, `v/vt/vn` triplets on every face corner as the reference's loader
requires (OBJ.cpp:66-72), one `g` + `usemtl` per object.

    python tools/blend_to_obj.py <file.blend> <out.obj> [--objects A,B] [--mtllib name.mtl]

A mesh without a UV layer (the bunny) gets planar texture coordinates from its bounding box.
This is offline tooling; the product never runs it.
"""
import argparse
import struct
import sys

import numpy as np


class Blend:
    def __init__(self, path):
        self.buf = open(path, "rb").read()
        head = self.buf[:12]
        if head[:7] != b"BLENDER":
            raise SystemExit(f"{path}: not an uncompressed .blend file")
        self.psz = 8 if head[7:8] == b"-" else 4
        self.end = "<" if head[8:9] == b"v" else ">"
        self.version = head[9:12].decode()
        self.blocks = []  # (code, size, old_ptr, sdna, count, offset)
        off = 12
        pf = "Q" if self.psz == 8 else "I"
        hs = 4 + 4 + self.psz + 4 + 4
        while off + hs <= len(self.buf):
            code = self.buf[off:off + 4]
            size, = struct.unpack_from(self.end + "i", self.buf, off + 4)
            old, = struct.unpack_from(self.end + pf, self.buf, off + 8)
            sdna, count = struct.unpack_from(self.end + "ii", self.buf, off + 8 + self.psz)
            self.blocks.append((code, size, old, sdna, count, off + hs))
            if code == b"ENDB":
                break
            off += hs + size
        self.by_ptr = {b[2]: b for b in self.blocks}
        self._parse_dna()

    def _parse_dna(self):
        blk = next(b for b in self.blocks if b[0] == b"DNA1")
        d = self.buf[blk[5]:blk[5] + blk[1]]
        assert d[:8] == b"SDNANAME"
        pos = 8

        def align():
            nonlocal pos
            pos = (pos + 3) & ~3

        def strings():
            nonlocal pos
            n, = struct.unpack_from(self.end + "i", d, pos)
            pos += 4
            out = []
            for _ in range(n):
                e = d.index(b"\0", pos)
                out.append(d[pos:e].decode())
                pos = e + 1
            align()
            return out

        self.names = strings()
        assert d[pos:pos + 4] == b"TYPE"
        pos += 4
        self.types = strings()
        assert d[pos:pos + 4] == b"TLEN"
        pos += 4
        self.tlen = list(struct.unpack_from(self.end + f"{len(self.types)}h", d, pos))
        pos += 2 * len(self.types)
        align()
        assert d[pos:pos + 4] == b"STRC"
        pos += 4
        n, = struct.unpack_from(self.end + "i", d, pos)
        pos += 4
        self.structs = []
        self.struct_of_type = {}
        for i in range(n):
            t, nf = struct.unpack_from(self.end + "hh", d, pos)
            pos += 4
            fields = []
            for _ in range(nf):
                ft, fn = struct.unpack_from(self.end + "hh", d, pos)
                pos += 4
                fields.append((self.types[ft], self.names[fn]))
            self.structs.append((self.types[t], fields))
            self.struct_of_type[self.types[t]] = i

    def field_size(self, tname, fname):
        n = 1
        base = fname
        while base.endswith("]"):
            l = base.rindex("[")
            n *= int(base[l + 1:-1])
            base = base[:l]
        if base.startswith("*") or base.startswith("(*"):
            return self.psz * n
        return self.tlen[self.types.index(tname)] * n

    def offsets(self, sname):
        _, fields = self.structs[self.struct_of_type[sname]]
        out, off = {}, 0
        for t, n in fields:
            key = n.lstrip("*").split("[")[0]
            out[key] = (off, t, n)
            off += self.field_size(t, n)
        return out

    def struct_blocks(self, sname):
        idx = self.struct_of_type[sname]
        return [b for b in self.blocks if b[3] == idx and b[0] != b"DNA1" and b[4] >= 1]

    def read(self, fmt, blk, off):
        return struct.unpack_from(self.end + fmt, self.buf, blk[5] + off)

    def ptr(self, blk, off):
        return self.read("Q" if self.psz == 8 else "I", blk, off)[0]


def mesh_of(bl, mesh_blk):
    mo = bl.offsets("Mesh")
    totvert, = bl.read("i", mesh_blk, mo["totvert"][0])
    totloop, = bl.read("i", mesh_blk, mo["totloop"][0])
    totpoly, = bl.read("i", mesh_blk, mo["totpoly"][0])

    def arr(field, dtype, count):
        p = bl.ptr(mesh_blk, mo[field][0])
        if p == 0 or p not in bl.by_ptr:
            return None
        b = bl.by_ptr[p]
        return np.frombuffer(bl.buf, dtype=dtype, count=count, offset=b[5])

    mvert = arr("mvert", np.dtype([("co", "<f4", 3), ("no", "<i2", 3), ("flag", "u1"), ("bw", "u1")]), totvert)
    mloop = arr("mloop", np.dtype([("v", "<i4"), ("e", "<i4")]), totloop)
    mpoly = arr("mpoly", np.dtype([("loopstart", "<i4"), ("totloop", "<i4"), ("mat", "<i2"), ("flag", "u1"),
                                   ("pad", "u1")]), totpoly)
    mloopuv = arr("mloopuv", np.dtype([("uv", "<f4", 2), ("flag", "<i4")]), totloop)
    return mvert, mloop, mpoly, mloopuv


def objects(bl):
    oo = bl.offsets("Object")
    io = bl.offsets("ID")
    out = []
    for b in bl.struct_blocks("Object"):
        name = bl.buf[b[5] + io["name"][0]: b[5] + io["name"][0] + 66].split(b"\0")[0].decode()[2:]
        data = bl.ptr(b, oo["data"][0])
        obmat = np.array(bl.read("16f", b, oo["obmat"][0]), dtype=np.float64).reshape(4, 4)  # [col][row]
        tgt = bl.by_ptr.get(data)
        if tgt is None or tgt[3] != bl.struct_of_type["Mesh"]:
            continue
        out.append((name, obmat, tgt))
    return out


def export(bl, out_path, want=None, mtllib=None, material=None):
    lines = [f"# reconstructed from a Blender {bl.version} file by tools/blend_to_obj.py"]
    if mtllib:
        lines.append(f"mtllib {mtllib}")
    vbase = tbase = nbase = 0
    ntri = 0
    for name, obmat, mesh_blk in objects(bl):
        if want and name not in want:
            continue
        mvert, mloop, mpoly, mloopuv = mesh_of(bl, mesh_blk)
        if mvert is None or mpoly is None:
            continue
        co = np.c_[mvert["co"].astype(np.float64), np.ones(len(mvert))] @ obmat  # row vector x [col][row]
        co = co[:, :3]
        nrm = mvert["no"].astype(np.float64) / 32767.0
        nrm = nrm @ np.linalg.inv(obmat[:3, :3]).T  # normals: inverse transpose
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
        # Blender -> OBJ axes: (x, y, z) -> (x, z, -y)
        co = np.c_[co[:, 0], co[:, 2], -co[:, 1]]
        nrm = np.c_[nrm[:, 0], nrm[:, 2], -nrm[:, 1]]
        if mloopuv is not None:
            uv = mloopuv["uv"].astype(np.float64)
        else:
            lo, hi = co.min(0), co.max(0)
            span = np.maximum(hi - lo, 1e-30)
            uv = (co[:, :2] - lo[:2]) / span[:2]  # planar map, one per vertex
        lines.append(f"o {name}")
        for p in co:
            lines.append(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}")
        for t in uv:
            lines.append(f"vt {t[0]:.6f} {t[1]:.6f}")
        for n in nrm:
            lines.append(f"vn {n[0]:.4f} {n[1]:.4f} {n[2]:.4f}")
        # `g` first and `usemtl` right after it: the reference's loader opens a group at `g` and
        # keeps the next `g` from merging into it only when f / mtllib / usemtl follows directly
        # (OBJ.cpp:93-106, 125-139)
        lines.append(f"g {name}")
        mat = material or "None"
        if isinstance(material, dict):
            mat = material.get(name, material.get("*", "None"))
        lines.append(f"usemtl {mat}")
        lines.append("s off")
        for pl in mpoly:
            ls, tl = int(pl["loopstart"]), int(pl["totloop"])
            per_loop_uv = mloopuv is not None
            corner = lambda l: (f"{vbase + int(mloop['v'][l]) + 1}/"
                                f"{tbase + (l if per_loop_uv else int(mloop['v'][l])) + 1}/"
                                f"{nbase + int(mloop['v'][l]) + 1}")
            for k in range(1, tl - 1):  # fan
                lines.append(f"f {corner(ls)} {corner(ls + k)} {corner(ls + k + 1)}")
                ntri += 1
        vbase += len(co)
        tbase += len(uv)
        nbase += len(nrm)
    # no trailing newline: the reference's `while(!f.eof())` loop re-reads the last token type
    # when the file ends in one (OBJ.cpp:29-31)
    open(out_path, "w").write("\n".join(lines))
    return ntri


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("blend")
    ap.add_argument("out")
    ap.add_argument("--objects", default="")
    ap.add_argument("--mtllib", default="")
    ap.add_argument("--material", default="", help="NAME for every object, or OBJECT=NAME,...,*=NAME")
    a = ap.parse_args()
    bl = Blend(a.blend)
    for name, _, m in objects(bl):
        mv, ml, mp, muv = mesh_of(bl, m)
        print(f"object {name}: {0 if mv is None else len(mv)} verts, {0 if mp is None else len(mp)} polys, "
              f"uv layer: {muv is not None}", file=sys.stderr)
    material = a.material or None
    if material and "=" in material:
        material = dict(kv.split("=", 1) for kv in material.split(","))
    n = export(bl, a.out, set(a.objects.split(",")) if a.objects else None, a.mtllib or None, material)
    print(f"{a.out}: {n} triangles", file=sys.stderr)
