"""Build an experimental variant of libprismer_hip.so next to the product library (for tools/ab_probe.py).

    python tools/build_variant.py NAME [file.hip:-DFLAG[,-DFLAG2]] ...      # working-tree sources + per-file flags
    python tools/build_variant.py NAME --rev HEAD~1                         # all sources taken from a git revision

Output: prismer_amd/lib/libprismer_hip_NAME.so (git-ignored).  Files without extra flags reuse the product build's objects.
"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prismer_amd import build as b

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def main():
    name = sys.argv[1]
    rest = sys.argv[2:]
    out = os.path.join(b.LIBDIR, f'libprismer_hip_{name}.so')
    tmp = tempfile.mkdtemp(prefix='phvar_')
    if rest and rest[0] == '--rev':
        rev = rest[1]
        src = os.path.join(tmp, 'a', 'b')                      # keeps "../../include/prismer_hip.h" resolvable
        os.makedirs(src); os.makedirs(os.path.join(tmp, 'include'))
        files = subprocess.check_output(['git', 'ls-tree', '--name-only', rev, 'prismer_amd/csrc/'], cwd=ROOT, text=True).split()
        for f in files:
            open(os.path.join(src, os.path.basename(f)), 'w').write(subprocess.check_output(['git', 'show', f'{rev}:{f}'], cwd=ROOT, text=True))
        open(os.path.join(tmp, 'include', 'prismer_hip.h'), 'w').write(
            subprocess.check_output(['git', 'show', f'{rev}:include/prismer_hip.h'], cwd=ROOT, text=True))
        jobs = [(s, os.path.join(src, s), []) for s in b.SOURCES]
    else:
        b.build(verbose=False)                                  # product objects are current
        extra = dict((r.split(':', 1)[0], r.split(':', 1)[1].split(',')) for r in rest)
        jobs = [(s, os.path.join(b.CSRC, s), extra.get(s)) for s in b.SOURCES]

    def cc(job):
        s, path, flags = job
        if flags is None:
            return os.path.join(b.LIBDIR, s.replace('.hip', '.o'))
        obj = os.path.join(tmp, s.replace('.hip', '.o'))
        r = subprocess.run([HIPCC] + b.FLAGS + flags + ['-c', path, '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f'hipcc failed on {s}:\n{r.stderr[-3000:]}')
        return obj

    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, jobs))
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit('link failed:\n' + r.stderr[-3000:])
    print('built', out)


if __name__ == '__main__':
    main()
