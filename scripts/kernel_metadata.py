"""Per-kernel code-object metadata of a built library: registers, spills, scratch, LDS (what the GPU actually runs).

    python scripts/kernel_metadata.py [path/to/lib.so] > profiles/rNN_kernel_metadata.txt

The fat binary section (.hip_fatbin) holds clang-offload-bundles; each gfx950 code object is an ELF whose notes carry the
amdhsa.kernels metadata (llvm-readelf --notes)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          're2nn-seq_amd', 'csrc', 'libfarnn_hip.so')
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, 'fatbin')
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fat], check=True)
    blob = open(fat, 'rb').read()
    rows = []
    # ELF code objects inside the bundles: split at ELF magics, let readelf tell which parse
    starts = [m.start() for m in re.finditer(b'\x7fELF\x02\x01\x01\x40', blob)]
    for k, st in enumerate(starts):
        co = os.path.join(tmp, 'co%d.elf' % k)
        open(co, 'wb').write(blob[st:(starts[k + 1] if k + 1 < len(starts) else len(blob))])
        r = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True)
        cur = {}
        for line in r.stdout.splitlines():
            m = re.match(r'\s+-?\s*\.(\w+):\s+(.*)', line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip().strip("'")
            if key == 'agpr_count' and cur.get('name'):
                pass
            if key in ('name', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size',
                       'group_segment_fixed_size', 'max_flat_workgroup_size'):
                if key == 'name' and val.startswith('_Z') is False and 'kernel' not in val:
                    continue
                cur[key] = val
            if key == 'wavefront_size' or (key == 'vgpr_spill_count'):
                if 'name' in cur and 'vgpr_count' in cur and 'vgpr_spill_count' in cur:
                    rows.append(cur)
                    cur = {}
    def dem(n):
        try:
            return subprocess.run([os.path.join(LLVM, 'llvm-cxxfilt'), n], capture_output=True, text=True).stdout.strip()
        except Exception:
            return n
    print('%-6s %-6s %-8s %-8s %-8s %s' % ('vgpr', 'sgpr', 'vspill', 'sspill', 'scratch', 'kernel'))
    for r in sorted(rows, key=lambda r: dem(r['name'])):
        print('%-6s %-6s %-8s %-8s %-8s %s' % (r.get('vgpr_count'), r.get('sgpr_count'), r.get('vgpr_spill_count'),
                                                r.get('sgpr_spill_count'), r.get('private_segment_fixed_size'), dem(r['name'])[:150]))
    print('# %d kernels; with scratch: %d' % (len(rows), sum(1 for r in rows if r.get('private_segment_fixed_size') not in (None, '0'))))
