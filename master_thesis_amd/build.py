"""Build libmt355.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmt355.so')
SOURCES = [('conv.hip', []), ('conv_halo.hip', []), ('conv_halo_wgrad.hip', []), ('conv_halo_wino.hip', []), ('conv_halo_wino25.hip', []), ('conv_halo_wino43.hip', []), ('conv_halo_planes.hip', []), ('conv_tiny.hip', []), ('stream.hip', ['-ffp-contract=off']),
           ('metrics.hip', []), ('api.cpp', []), ('comm.cpp', []), ('dispatch.cpp', [])]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_common.h'), os.path.join(CSRC, 'dispatch.h'),
               os.path.join(ROOT, 'include', 'mt355.h')]
    objs, cmds = [], []
    for src, extra in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            cmds.append([hipcc, '-x', 'hip', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-fvisibility=hidden',
                         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-c', path, '-o', obj] + extra +
                        os.environ.get('MT355_CFLAGS', '').split() +                       # lab builds: extra flags for all /
                        os.environ.get('MT355_CFLAGS_' + os.path.splitext(src)[0], '').split())   # one translation unit
    if cmds:      # the translation units compile side by side (conv_halo.hip alone is ~1.5 minutes)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(8, len(cmds))) as ex:
            list(ex.map(run, cmds))
    if force or _stale(LIB, objs + [os.path.join(CSRC, 'exports.map')]):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + \
            ['-ldl', '-Wl,--version-script=' + os.path.join(CSRC, 'exports.map')]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
