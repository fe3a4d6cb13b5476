#!/usr/bin/env python3
"""Link-time guard for the hand-written loop of chain_dp_tile (csrc/chain_dp_tile.h).

The assembly addresses the kernel's LDS from byte 0 with immediate offsets (struct Lds<NX, NF, GEN, TAB, C16>), which is only right while the
kernel owns exactly ONE LDS object.  A second `__shared__` object in that kernel would be laid out beside it and the group segment of the
kernel would grow by its size, so: for every chain_dp_tile instantiation in the library's gfx950 code objects, the
`.group_segment_fixed_size` of the kernel descriptor must equal Lds<>::BYTES computed from the template arguments in the kernel's name.

Usage: check_lds_layout.py path/to/libmm2chain_hip.so      (exit 0 = all instantiations agree; prints one line per kernel with -v)
The Makefile runs it after linking; tests/test_cpu_abi.py runs it on the shipped library.  Needs only Python + msgpack (code-object metadata).
"""
import re
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def lds_bytes(nx, nf, gen, tab, ring):
    """Lds<NX, NF, GEN, TAB, RING>::BYTES of csrc/chain_dp_tile.h: x / q ring (NX tiles of 64 slots of 8 bytes; 4 in the compact form, RING 1, and in the
    q24 form, RING 2, which adds one byte per ring anchor for bits 16-23 of q), f / p ring (NF tiles), one stamp byte per ring anchor, the gap-cost
    table, the segment-id ring"""
    return nx * 64 * (4 if ring else 8) + 2 * nf * 256 + 64 * nx + (64 * nx if ring == 2 else 0) + (1024 if tab else 0) + (64 * nx if gen else 0)


def elf_sections(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not an ELF64"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    str_off = secs[shstrndx][2]
    out = []
    for name, typ, off, size in secs:
        end = elf.index(b"\0", str_off + name)
        out.append((elf[str_off + name:end].decode(), typ, off, size))
    return out


def kernels_of(elf):
    """[(name, group_segment_fixed_size)] from the NT_AMDGPU_METADATA note (type 32, msgpack)"""
    for _name, typ, off, size in elf_sections(elf):
        if typ != 7:                                   # SHT_NOTE
            continue
        at = off
        while at < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, at)
            at += 12
            nname = elf[at:at + namesz].rstrip(b"\0")
            at += (namesz + 3) & ~3
            desc = elf[at:at + descsz]
            at += (descsz + 3) & ~3
            if nname == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False)
                return [(k[".name"], int(k[".group_segment_fixed_size"])) for k in md.get("amdhsa.kernels", [])]
    return []


def code_objects(so_bytes):
    """every amdgcn ELF bundled in the file (one bundle per device translation unit)"""
    for m in re.finditer(re.escape(MAGIC), so_bytes):
        base = m.start()
        n, = struct.unpack_from("<Q", so_bytes, base + len(MAGIC))
        at = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", so_bytes, at)
            triple = so_bytes[at + 24:at + 24 + tlen].decode()
            at += 24 + tlen
            if "amdgcn" in triple and size > 0:
                yield triple, so_bytes[base + off:base + off + size]


def check(path, verbose=False):
    data = open(path, "rb").read()
    seen, bad = 0, []
    for triple, elf in code_objects(data):
        for name, lds in kernels_of(elf):
            # _ZN4mm2c13chain_dp_tileILi8ELi2ELb1ELb0ELb1ELb1ELb0ELi1EEEv...: <NX, NF, SKIP, GEN, GS1, FAR, TAB, RING>
            m = re.match(r"_ZN4mm2c13chain_dp_tileILi(\d+)ELi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])ELb([01])ELi([012])EEE", name)
            mc = re.match(r"_ZN4mm2c13chain_dp_coopILi(\d+)ELb([01])ELb([01])ELb([01])EEE", name)
            if mc:
                # chain_dp_coop<W, GS1, FAR, TAB> (csrc/chain_dp_coop.h): the same hand-written loop over Lds<COOP_NX = 16, COOP_NF = 8, false, TAB, false>, with the
                # per-anchor summaries (two sets of a 64-bit key and a count per lane) and two tiles' pair tables (64 x 64 ints each) behind the rings INSIDE the one LDS object
                want = lds_bytes(16, 8, 0, int(mc.group(4)), 0) + 2 * 64 * 8 + 2 * 64 * 4 + 2 * 2 * 64 * 8 + 2 * 64 * 64 * 4 + 32 * 64 * 12 + 16   # ... and (round 6) the candidate rings: x / q and f of 32 tiles, and the group counters of phase A1
                seen += 1
                if verbose:
                    print(f"{triple} chain_dp_coop<{','.join(mc.groups())}>: group segment {lds} B, rings + summaries {want} B")
                if lds != want:
                    bad.append((name, lds, want))
                continue
            if not m:
                continue
            nx, nf, _skip, gen, _gs1, _far, tab, c16 = (int(v) for v in m.groups())
            want = lds_bytes(nx, nf, gen, tab, c16)
            seen += 1
            if verbose:
                print(f"{triple} chain_dp_tile<{','.join(m.groups())}>: group segment {lds} B, Lds<>::BYTES {want} B")
            if lds != want:
                bad.append((name, lds, want))
    if seen == 0:
        raise SystemExit(f"{path}: no chain_dp_tile kernel found in its code objects")
    for name, lds, want in bad:
        print(f"{name}: group segment is {lds} B but the kernel's one LDS object is {want} B -- the hand-written loop assumes it sits at LDS offset 0", file=sys.stderr)
    return seen, bad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "-v"]
    seen, bad = check(args[0], verbose="-v" in sys.argv)
    if bad:
        raise SystemExit(1)
    print(f"{args[0]}: {seen} chain_dp_tile instantiations, each owns exactly its Lds<> object")
