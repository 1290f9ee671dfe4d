"""Kernel resource metadata of the built library, read from the file (no GPU, no tools).

`libamdkge.so` carries one clang offload bundle per translation unit in `.hip_fatbin`; each holds the gfx950 code
object (an ELF) whose `NT_AMDGPU_METADATA` note is a msgpack map with one entry per kernel: `.vgpr_count`,
`.sgpr_count`, `.private_segment_fixed_size` (scratch bytes per lane), `.group_segment_fixed_size` (static LDS).
`kernel_resources()` returns them keyed by the kernels' mangled names -- what `tests/test_kernel_resources.py`
holds the hot kernels' occupancy to (a register more than 168 takes the headline forward kernel from three waves
per SIMD to two: +13 % on the step, found the hard way in round 4).
"""
from __future__ import annotations

import struct

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundles(blob: bytes):
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, cur)
            cur += 24
            triple = blob[cur:cur + tl].decode()
            cur += tl
            yield triple, blob[pos + off: pos + off + size]
        pos = blob.find(MAGIC, pos + 1)


def _notes(elf: bytes):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        base = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, base + 4)
        if sh_type != 7:   # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, base + 0x18)
        cur, end = off, off + size
        while cur + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, cur)
            cur += 12
            name = elf[cur:cur + namesz].rstrip(b"\0")
            cur += (namesz + 3) & ~3
            desc = elf[cur:cur + descsz]
            cur += (descsz + 3) & ~3
            yield name, ntype, desc


def kernel_resources(path: str, arch: str = "gfx950") -> dict:
    """{mangled kernel name: {"vgpr", "agpr", "sgpr", "scratch", "lds", "waves_per_simd"}} for every kernel in `path`."""
    import msgpack
    blob = open(path, "rb").read()
    out = {}
    for triple, obj in _bundles(blob):
        if arch not in triple or not obj:
            continue
        for name, ntype, desc in _notes(obj):
            if name != b"AMDGPU" or ntype != 32:   # NT_AMDGPU_METADATA
                continue
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in meta.get("amdhsa.kernels", []):
                v, a = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
                # gfx950: 512 unified registers per SIMD lane, allocated in blocks of 8, at most 8 waves;
                # `.vgpr_count` is the unified total (it includes the accumulation registers)
                total = max(8, ((v + 7) // 8) * 8)
                out[k[".name"]] = dict(vgpr=v, agpr=a, sgpr=int(k.get(".sgpr_count", 0)),
                                       scratch=int(k.get(".private_segment_fixed_size", 0)),
                                       lds=int(k.get(".group_segment_fixed_size", 0)),
                                       waves_per_simd=min(8, 512 // total))
    return out
