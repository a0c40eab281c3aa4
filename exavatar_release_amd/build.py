"""Builds the in-tree gfx950 shared library ``exavatar_release_amd/libexa_raster.so`` with hipcc.

Plain ``hipcc --offload-arch=gfx950`` on the ``.hip`` sources under ``csrc/``; no cmake, no torch
extension machinery, no torch headers (the library is a pure C ABI, include/exa_raster.h).
The ``.so`` stays in-tree so it travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libexa_raster.so')
BUILD = os.path.join(HERE, '_build')

ARCH = 'gfx950'
# -fno-slp-vectorize: left to itself the compiler packs pairs of independent scalar float operations of the blends into
# v_pk_*_f32 -- which take two passes of the VALU like the two scalar ones, i.e. save nothing where the kernel is bound by
# VALU throughput -- and pays for it with v_mov shuffles into and out of register pairs (render_fwd: 64 moves and 66 packed
# operations against 54 and 52).  The packed operations written out in blend.h (ext_vector_type) stay.  Same arithmetic, same
# bits; render_bwd 44.1 -> 42.1 us by events, the C3 step 142.1 -> 139.6 us (7 023 / 7 052 -> 7 167 / 7 164 it/s interleaved).
COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-fno-slp-vectorize',
          '-Wall', '-Wno-unused-function'] + (['-DEXA_PROBE_SORT'] if os.environ.get('EXA_PROBE_SORT') else []) + \
    (['-DEXA_PROBE_FWD'] if os.environ.get('EXA_PROBE_FWD') else [])
# per-file extra flags: the forward per-Gaussian stage is the bit-exact-with-oracle part
SOURCES = {
    'preprocess_fwd.hip': ['-ffp-contract=off'],
    'binning.hip': [],
    'render_fwd.hip': [],
    'render_bwd.hip': [],
    'compose.hip': [],
    'preprocess_bwd.hip': [],
    'ssim.hip': [],
    'api.hip': [],
}


def hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


BINDING_SRC = 'torch_binding.cpp'        # the compiled autograd node (host code over the C ABI): its own artefact and stamp
BINDING = os.path.join(HERE, '_exa_torch.so')


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name == BINDING_SRC:
            continue
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(name.encode())
            h.update(f.read())
    with open(os.path.join(HERE, '..', 'include', 'exa_raster.h'), 'rb') as f:
        h.update(f.read())
    h.update(repr((COMMON, SOURCES)).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile (if sources changed) and return the library path."""
    os.makedirs(BUILD, exist_ok=True)
    stamp = os.path.join(BUILD, 'stamp')
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    cc = hipcc()
    objs = []
    for src, extra in SOURCES.items():
        obj = os.path.join(BUILD, src.replace('.hip', '.o'))
        cmd = [cc] + COMMON + extra + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [cc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(stamp, 'w') as f:
        f.write(dig)
    return LIB


def _binding_digest():
    import torch
    h = hashlib.sha256()
    for path in (os.path.join(CSRC, BINDING_SRC), os.path.join(HERE, '..', 'include', 'exa_raster.h')):
        with open(path, 'rb') as f:
            h.update(f.read())
    h.update(repr((torch.__version__, BINDING_FLAGS)).encode())
    return h.hexdigest()


BINDING_FLAGS = ['-O2', '-std=c++17', '-fPIC', '-shared', '-w', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
                 '-DTORCH_EXTENSION_NAME=_exa_torch', '-DTORCH_API_INCLUDE_EXTENSION_H', '-D_GLIBCXX_USE_CXX11_ABI=1']


def build_binding(force=False, verbose=False):
    """Compile ``csrc/torch_binding.cpp`` -- the compiled autograd node of the drop-in surface: host C++ against torch's
    headers and ``include/exa_raster.h``, no device code -- into the in-tree Python extension ``_exa_torch.so`` (g++, ~40 s
    once; it opens ``libexa_raster.so`` itself at ``init``).  Returns its path."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(BUILD, exist_ok=True)
    stamp = os.path.join(BUILD, 'stamp_binding')
    dig = _binding_digest()
    if not force and os.path.exists(BINDING) and os.path.exists(stamp) and open(stamp).read() == dig:
        return BINDING
    cxx = os.environ.get('CXX') or shutil.which('g++') or shutil.which('c++')
    if not cxx:
        raise RuntimeError('g++ not found')
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    cmd = [cxx] + BINDING_FLAGS + ['-I' + p for p in ce.include_paths()] + \
        ['-I' + sysconfig.get_paths()['include'], '-I' + os.path.join(rocm, 'include'), os.path.join(CSRC, BINDING_SRC),
         '-o', BINDING, '-L' + tlib, '-ltorch', '-ltorch_cpu', '-ltorch_python', '-lc10', '-lc10_hip', '-lamdhip64', '-ldl',
         '-Wl,-rpath,' + tlib]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(stamp, 'w') as f:
        f.write(dig)
    return BINDING


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_binding(force='--force' in sys.argv, verbose=True))
