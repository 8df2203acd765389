"""Build the HIP library in-tree:  python -m ga_gan_amd.build  (hipcc, gfx950 only)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, 'libsgops.so')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(HERE, 'csrc', '*.hip')))


def is_stale():
    if not os.path.isfile(OUT):
        return True
    deps = sources() + glob.glob(os.path.join(HERE, 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h'))
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=True):
    if not force and not is_stale():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, *FLAGS, '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(HERE, 'csrc'), *sources(), '-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
