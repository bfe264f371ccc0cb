"""Build libfarnn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python re2nn-seq_amd/csrc/build.py [--force] [--probes]

Every `*.hip` file of this directory is one translation unit: they compile in parallel to
`build/<name>.o` and link into `libfarnn_hip.so`.  A unit is recompiled when the sha256 of its
source, of the headers it included last time (hipcc -MD) or of the flags changed -- content, not
mtimes, so a prebuilt library newer than edited sources is never mistaken for current.
`--probes` (or FARNN_PROBES=1) builds the profiling variant libfarnn_hip_probes.so beside it (objects
under build_probes/): -DFARNN_PROBES compiles the kernels' s_memtime phase probes and their printf
in; the production library has none of them.  FARNN_LIB=<path> makes the package load that variant.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'libfarnn_hip.so')
OBJ = os.path.join(HERE, 'build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def _units():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.hip'))


def _flags(probes):
    extra = os.environ.get('FARNN_EXTRA_FLAGS', '').split()
    # the profiling build is also the A/B build: it carries the forms the production library left behind (FARNN_AB: the one-launch CRF
    # step, chain_viterbi*.hip) beside the in-kernel probes
    return FLAGS + (['-DFARNN_PROBES=1', '-DFARNN_AB=1'] if probes else []) + extra


def _deps(depfile):
    """the prerequisites a make-style depfile lists (system headers under /opt/rocm are skipped)"""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace('\\\n', ' ')
    deps = []
    for line in text.split('\n'):
        if ':' not in line:
            continue
        for tok in line.split(':', 1)[1].split():
            if not tok.startswith(('/opt/', '/usr/')):
                deps.append(tok if os.path.isabs(tok) else os.path.join(HERE, tok))
    return deps


def _digest(files, flags):
    h = hashlib.sha256(' '.join(flags).encode())
    for f in sorted(set(files)):
        try:
            h.update(open(f, 'rb').read())
        except OSError:
            return None
    return h.hexdigest()


def _unit_flags(src):
    """extra flags a unit asks for itself: a line `// build-flags: <flags>` in its first lines"""
    with open(src) as f:
        for _, line in zip(range(12), f):
            if line.startswith('// build-flags:'):
                return line.split(':', 1)[1].split()
    return []


def _unit_checks(src):
    """ISA checks a unit asks for: lines `// build-check: ring-registers <kernel-name substring>` in its first lines.  The unit is
    compiled with -save-temps=obj and scripts/check_ring_registers.py walks the device assembly of the matching kernels: no
    compiler-generated copy or spill may read a register whose asm-issued global load is still in flight (the kernels that keep a
    ring of loads in ordinary asm outputs: chain_dest.hip.h).  A finding fails the build."""
    out = []
    with open(src) as f:
        for _, line in zip(range(16), f):
            if line.startswith('// build-check: ring-registers'):
                out.append(line.split('ring-registers', 1)[1].strip())
    return out


def _run_checks(src, objdir, checks):
    name = os.path.splitext(os.path.basename(src))[0]
    asm = os.path.join(objdir, name + '-hip-amdgcn-amd-amdhsa-gfx950.s')
    script = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'scripts', 'check_ring_registers.py')
    for pat in checks:
        r = subprocess.run([sys.executable, script, asm, '--kernels', pat], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('ring-register check failed for {} ({}):\n{}'.format(name, pat, (r.stdout + r.stderr)[-3000:]))


def _compile(src, flags, force, verbose, objdir, run_checks=True):
    flags = flags + _unit_flags(src)
    # (the shipped library only: the profiling build's device printf re-uses the ring's registers behind the loops, on paths the
    #  check's model cannot tell from a live ring)
    checks = _unit_checks(src) if run_checks else []
    if checks:
        flags = flags + ['-save-temps=obj']
    name = os.path.splitext(os.path.basename(src))[0]
    obj, dep, stamp = (os.path.join(objdir, name + ext) for ext in ('.o', '.d', '.sha'))
    if not force and os.path.exists(obj):
        deps = _deps(dep)
        if deps is not None:
            want = _digest([src] + deps, flags)
            try:
                if want and open(stamp).read().strip() == want:
                    return obj, False
            except OSError:
                pass
    cmd = [HIPCC] + flags + ['-c', '-MD', '-MF', dep, '-o', obj, src]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    if checks:
        _run_checks(src, objdir, checks)
    digest = _digest([src] + (_deps(dep) or []), flags)
    with open(stamp, 'w') as f:
        f.write(digest or '')
    return obj, True


def build_hip(force=False, verbose=True, probes=None):
    if probes is None:
        probes = os.environ.get('FARNN_PROBES', '') not in ('', '0')
    objdir = OBJ + '_probes' if probes else OBJ
    out = OUT.replace('.so', '_probes.so') if probes else OUT
    os.makedirs(objdir, exist_ok=True)
    flags = _flags(probes)
    units = _units()
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as pool:
        done = list(pool.map(lambda s: _compile(s, flags, force, verbose, objdir, run_checks=not probes), units))
    objs = [o for o, _ in done]
    # the link has a stamp of its own: the object list, every unit's digest and the link flags.  A link that failed or was
    # interrupted after the units were stamped, or a unit that was deleted / renamed, leaves a stamp that no longer matches --
    # the stale library is never returned as current
    link_cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs
    h = hashlib.sha256(' '.join(link_cmd).encode())
    for o in objs:
        try:
            h.update(open(os.path.splitext(o)[0] + '.sha', 'rb').read())
        except OSError:
            h.update(b'?')
    want = h.hexdigest()
    stamp = os.path.join(objdir, 'link.sha')
    try:
        have = open(stamp).read().strip()
    except OSError:
        have = ''
    if any(changed for _, changed in done) or not os.path.exists(out) or have != want:
        if os.path.exists(stamp):
            os.remove(stamp)
        if os.path.exists(out):
            os.remove(out)
        if verbose:
            print(' '.join(link_cmd), flush=True)
        subprocess.run(link_cmd, check=True, cwd=HERE)
        with open(stamp, 'w') as f:
            f.write(want)
    return out


def build_rccl(force=False, verbose=True):
    """libfarnn_rccl.so (include/farnn_rccl.h): the tag gather over RCCL as a C-ABI of its own; host code only, links librccl.so.
    A library of its own so that libfarnn_hip.so does not depend on RCCL."""
    src = os.path.join(HERE, 'rccl', 'farnn_rccl.cpp')
    out = os.path.join(HERE, 'libfarnn_rccl.so')
    hdr = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'farnn_rccl.h')
    cmd = [HIPCC, '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__', '-x', 'c++', src, '-o', out, '-I/opt/rocm/include',
           '-L/opt/rocm/lib', '-lrccl', '-lamdhip64', '-Wl,-rpath,/opt/rocm/lib']
    want = _digest([src, hdr], cmd)
    stamp = os.path.join(OBJ, 'rccl.sha')
    os.makedirs(OBJ, exist_ok=True)
    try:
        have = open(stamp).read().strip()
    except OSError:
        have = ''
    if force or not os.path.exists(out) or have != want:
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
        with open(stamp, 'w') as f:
            f.write(want or '')
    return out


if __name__ == '__main__':
    build_hip(force='--force' in sys.argv, probes=True if '--probes' in sys.argv else None)
    build_rccl(force='--force' in sys.argv)
