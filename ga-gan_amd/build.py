"""Build the HIP library in-tree:  python -m ga_gan_amd.build  (hipcc, gfx950 only).

Every csrc/*.hip is compiled to an object of its own (in parallel) and the objects are linked into
libsgops.so.  Staleness is decided by content, not by time stamps: each object carries a stamp with the
digest of its source, of every header and of the compiler flags, the way the reference keys its plugin build
directory by a digest of the op's sources (torch_utils/custom_ops.py:88-104).  The whole build runs under an
exclusive file lock, so that several ranks importing a stale tree at once build it once and the others wait
(the reference's FileBaton, torch_utils/custom_ops.py:105-115).
"""
import concurrent.futures
import fcntl
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, 'libsgops.so')
OBJ_DIR = os.path.join(HERE, 'build')
CFLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wall', '-Wno-unused-function',
          '-Wno-unused-const-variable']
# per-source additions.  The epilogues of the band and one-pass up-sampling kernels run beside matrix instructions, where packed float32 vector
# instructions (what the SLP vectoriser makes of adjacent scalar multiplies) cost more than the two scalar ones.
EXTRA_CFLAGS = {'sg_band3x3': ['-fno-slp-vectorize'], 'sg_convup_fir': ['-fno-slp-vectorize']}


def sources():
    return sorted(glob.glob(os.path.join(HERE, 'csrc', '*.hip')))


def headers():
    return sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'include', '*.h')))


def _digest(paths, extra=''):
    h = hashlib.md5(extra.encode())
    for path in paths:
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _plan():
    """[(source, object, stamp path, wanted digest)] and the digest the linked library must carry."""
    common = _digest(headers(), ' '.join(CFLAGS))
    items = []
    for src in sources():
        stem = os.path.splitext(os.path.basename(src))[0]
        items.append((src, os.path.join(OBJ_DIR, stem + '.o'), os.path.join(OBJ_DIR, stem + '.md5'),
                      _digest([src], common + ' '.join(EXTRA_CFLAGS.get(stem, [])))))
    return items, hashlib.md5(''.join(i[3] for i in items).encode()).hexdigest()


def is_stale():
    _, want = _plan()
    return not os.path.isfile(OUT) or _read(os.path.join(OBJ_DIR, 'libsgops.md5')) != want


def build(force=False, verbose=True, jobs=None):
    if not force and not is_stale():
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)              # one builder per tree; the others wait here ...
        try:
            if not force and not is_stale():          # ... and find the work done
                return OUT
            hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
            include = ['-I', os.path.join(ROOT, 'include'), '-I', os.path.join(HERE, 'csrc')]
            items, want = _plan()
            todo = [i for i in items if force or not os.path.isfile(i[1]) or _read(i[2]) != i[3]]

            def compile_one(item):
                src, obj, stamp, digest = item
                stem = os.path.splitext(os.path.basename(src))[0]
                cmd = [hipcc, *CFLAGS, *EXTRA_CFLAGS.get(stem, []), *include, '-c', src, '-o', obj]
                if verbose:
                    print(' '.join(cmd), flush=True)
                subprocess.run(cmd, check=True)
                with open(stamp, 'w') as f:
                    f.write(digest)

            jobs = jobs or min(len(todo) or 1, int(os.environ.get('MAX_JOBS', 0)) or os.cpu_count() or 1, 16)
            with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
                list(pool.map(compile_one, todo))
            tmp = OUT + f'.tmp{os.getpid()}'
            cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *[i[1] for i in items], '-o', tmp]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            os.replace(tmp, OUT)                      # a process that has the old library mapped keeps its copy
            with open(os.path.join(OBJ_DIR, 'libsgops.md5'), 'w') as f:
                f.write(want)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
