#!/usr/bin/env python3
"""per-kernel resources of a built library (CPU; no GPU needed): VGPRs, spills, scratch, static LDS of every gfx950 kernel in its embedded code
objects, read from the code-object metadata with llvm-readelf.  usage: python tools/code_objects.py [lib.so] [--spills]
tests/test_code_objects.py holds the kernels with counted `s_waitcnt vmcnt(N)` schedules to zero spills (scratch traffic would join the count)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
FIELDS = ('vgpr_count', 'agpr_count', 'vgpr_spill_count', 'sgpr_count', 'sgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size')


def code_objects(path):
    """the gfx950 ELF images of every clang offload bundle embedded in the host library"""
    d = open(path, 'rb').read()
    out = []
    for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', d):
        p = m.start() + 24
        n, = struct.unpack_from('<Q', d, p)
        p += 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', d, p)
            p += 24
            triple = d[p:p + tl].decode()
            p += tl
            if 'gfx950' in triple and size:
                out.append(d[m.start() + off:m.start() + off + size])
    return out


def kernels(path):
    """{mangled kernel name: {field: int}}"""
    res = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(co)
            name = f.name
        try:
            txt = subprocess.run([READELF, '--notes', name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(name)
        for rec in re.split(r'\n\s+- \.agpr_count', '\n' + txt):   # one metadata record per kernel, keys in alphabetical order
            m = re.search(r'\.name:\s+(\S+)', rec)
            if not m:
                continue
            cur = {}
            for k in FIELDS:
                mm = re.search((r'^:\s+(\d+)' if k == 'agpr_count' else r'\.' + k + r':\s+(\d+)'), rec, re.M)
                cur[k] = int(mm.group(1)) if mm else 0
            res[m.group(1)] = cur
    return res


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    lib = args[0] if args else os.path.join(ROOT, 'ecg-representation-learning_amd', 'libecgvit_hip.so')
    ks = kernels(lib)
    for n, k in sorted(ks.items()):
        if '--spills' in sys.argv and not (k['vgpr_spill_count'] or k['private_segment_fixed_size']):
            continue
        print(f"{n[:120]:120s} vgpr {k['vgpr_count']:3d} agpr {k['agpr_count']:3d} spill {k['vgpr_spill_count']:3d} scratch {k['private_segment_fixed_size']:4d} B  lds {k['group_segment_fixed_size']:6d} B")
    print(len(ks), 'kernels in', lib)
