#!/usr/bin/env python
"""Per-kernel register / spill / LDS report of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_regs.py slotformer_amd/csrc/layer_fused.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/tmp/_kr.o',
                    '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
    if not m:
        if 'error' in line:
            print(line)
        continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        name = t.split(':', 1)[1].strip()
        try:
            name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        cur = {'name': name}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for c in rows:
    if flt and flt not in c['name']:
        continue
    print(f"{c['name'][:70]:70s} VGPR {c.get('VGPRs','?'):>4} AGPR {c.get('AGPRs','?'):>3} spill {c.get('VGPRs Spill','?'):>3} "
          f"scratch {c.get('ScratchSize [bytes/lane]','?'):>4} occ {c.get('Occupancy [waves/SIMD]','?')} LDS {c.get('LDS Size [bytes/block]','?')}")
