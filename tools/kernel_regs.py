#!/usr/bin/env python3
"""registers, spills and LDS of the kernels in a library's gfx950 code objects whose name contains a pattern: python tools/kernel_regs.py lib.so coop"""
import struct
import sys

import msgpack

sys.path.insert(0, __import__("os").path.dirname(__file__))
import check_lds_layout as c

so = open(sys.argv[1], "rb").read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for triple, elf in c.code_objects(so):
    for _n, typ, off, size in c.elf_sections(elf):
        if typ != 7:
            continue
        at = off
        while at < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, at); at += 12
            nname = elf[at:at + namesz].rstrip(b"\0"); at += (namesz + 3) & ~3
            desc = elf[at:at + descsz]; at += (descsz + 3) & ~3
            if nname == b"AMDGPU" and ntype == 32:
                for k in msgpack.unpackb(desc, raw=False).get("amdhsa.kernels", []):
                    if pat in k[".name"]:
                        print(k[".name"][:70], "vgpr", k[".vgpr_count"], "sgpr", k[".sgpr_count"], "spill v", k.get(".vgpr_spill_count"), "s", k.get(".sgpr_spill_count"),
                              "lds", k[".group_segment_fixed_size"], "scratch", k[".private_segment_fixed_size"])
