#!/usr/bin/env python3
"""Static per-kernel statistics of the gfx950 ISA of a .hip file (registers, scratch, LDS, occupancy, instruction mix,
spill traffic through VGPR lanes) -- the table DESIGN.md quotes.  Usage: tools/isa_stats.py [file.hip] [name-filter ...]

    hipcc -S --cuda-device-only with the flags of oflibpytorch_amd/_build.py, then a scan of the assembly text."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oflibpytorch_amd import _build  # noqa: E402


def main():
    src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.hip') else os.path.join(ROOT, 'oflibpytorch_amd/csrc/ofl_kernels.hip')
    filters = [a for a in sys.argv[1:] if not a.endswith('.hip')]
    flags = [f for f in _build.HIPCC_FLAGS if f not in ('-shared', '-fPIC')]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run([_build.hipcc_path()] + flags + ['-I', os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-o', out, src],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    demangle = lambda n: subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    print("%-96s %5s %5s %6s %6s %4s %6s %6s %6s %6s" % ("kernel", "vgpr", "sgpr", "scr B", "LDS B", "occ", "VALU", "SALU", "lane", "scr op"))
    for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\.Lfunc_end\d+:', text, flags=re.M | re.S):
        name, body = m.group(1), m.group(2)
        if 'amdhsa_kernel ' + name not in text:
            continue
        pretty = re.sub(r'\(anonymous namespace\)::', '', demangle(name))
        pretty = re.sub(r'\(.*$', '', pretty).replace('void ', '')
        if filters and not any(f in pretty for f in filters):
            continue
        meta = text[text.index('.amdhsa_kernel ' + name):]
        meta = meta[:meta.index('.end_amdhsa_kernel')]
        tail = text[m.end():m.end() + 6000]
        g = lambda k: (re.search(r'; ' + k + r': (\d+)', tail) or [None, '?'])[1]
        print("%-96s %5s %5s %6s %6s %4s %6d %6d %6d %6d" % (
            pretty[:96], g('NumVgprs'), g('NumSgprs'), g('ScratchSize'), g('LDSByteSize'), g('Occupancy'),
            len(re.findall(r'\n\tv_', body)), len(re.findall(r'\n\ts_', body)),
            len(re.findall(r'v_readlane|v_writelane', body)), len(re.findall(r'\tscratch_', body))))


if __name__ == '__main__':
    main()
