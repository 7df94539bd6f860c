"""
Build libltmi.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m libertem_amd.build [--force]
    python -m libertem_amd.build --hardened  # host side with checked std:: containers ->
                                             # libertem_amd/_lib/libltmi_hardened.so
    python -m libertem_amd.build --asan      # host side instrumented with AddressSanitizer ->
                                             # libertem_amd/_lib/libltmi_asan.so

hipcc cross-compiles without a GPU.  The result lands in libertem_amd/_lib/libltmi.so, which is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
import shutil
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, '_lib')
LIB = os.path.join(LIBDIR, 'libltmi.so')
OBJDIR = os.path.join(HERE, '_lib', 'obj')

SOURCES = ['ltmi_capi.cpp', 'ltmi_comm.cpp', 'ltmi_dense.hip', 'ltmi_sparse.hip', 'ltmi_reduce.hip', 'ltmi_fft.hip', 'ltmi_dense64.hip', 'ltmi_bell.hip', 'ltmi_mib.hip', 'ltmi_split.hip', 'ltmi_scatter.hip', 'ltmi_cryst.hip', 'ltmi_fold.hip', 'ltmi_guard.hip']
# hipFFT for the Fourier-space operators (ltmi_fft.hip).  The loader binds libhipfft.so.0 to the copy
# torch already has in the process (same soname), i.e. the one that matches torch's HIP runtime.
LINK_LIBS = ['-L/opt/rocm/lib', '-lhipfft', '-ldl']
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function', '-Wno-inline-asm',
         '-fvisibility=hidden']   # exported: what include/ltmi.h declares, nothing else


def find_hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _deps(src):
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, 'ltmi_common.h'), header_path(),
            os.path.join(CSRC, 'ltmi_scatter_loop.inc') if src == 'ltmi_scatter.hip' else '',
            os.path.abspath(__file__)]
    return [d for d in deps if os.path.exists(d)]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def check_generated():
    """csrc/ltmi_scatter_loop.inc is generated (scripts/gen_scatter_asm.py) and tracked: a build from a tree in which
    the two disagree -- the generator edited, the file not regenerated, or the file edited by hand -- fails.
    (An installed package without scripts/ has nothing to compare with.)"""
    gen = os.path.join(os.path.dirname(HERE), 'scripts', 'gen_scatter_asm.py')
    inc = os.path.join(CSRC, 'ltmi_scatter_loop.inc')
    if not os.path.exists(gen):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location('_gen_scatter_asm', gen)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(inc) as f:
        have = f.read()
    if have != mod.render():
        raise RuntimeError(f"{inc} is not what {gen} generates: run `python scripts/gen_scatter_asm.py`")


def build(force=False, verbose=True, asan=False, hardened=False):
    """Sanitizer builds of the HOST code of the library (image builders, argument checks, the C ABI;
    device code unchanged), selected at run time with LTMI_LIB=<path>:

    hardened=True -> libltmi_hardened.so: -D_GLIBCXX_ASSERTIONS (every std::vector / std::string
        index of the image builders is bounds-checked and aborts with a message) + stack protector;
        no sanitizer runtime, so it coexists with the stock torch wheel (host-side -fsanitize=undefined
        was tried too: hipcc then mis-launches the unaligned-row kernel variants, so it is left out):
            LTMI_LIB=libertem_amd/_lib/libltmi_hardened.so python -m pytest tests -m gpu
    asan=True -> libltmi_asan.so (-fsanitize=address).  Needs an ASAN-enabled ROCm stack: ROCm's ASAN
        runtime intercepts the HSA allocator and aborts inside the un-instrumented HIP runtime that the
        torch wheel bundles (observed on this image), so with stock torch use the hardened build.
            LTMI_LIB=.../libltmi_asan.so LD_PRELOAD=$(clang -print-file-name=libclang_rt.asan-x86_64.so)"""
    hipcc = find_hipcc()
    check_generated()
    tag = '_asan' if asan else ('_hardened' if hardened else '')
    objdir = OBJDIR + tag
    lib = LIB.replace('libltmi.so', f'libltmi{tag}.so')
    extra = []
    if asan:
        extra = ['-fsanitize=address', '-fno-omit-frame-pointer', '-g', '-shared-libsan']
    elif hardened:
        extra = ['-D_GLIBCXX_ASSERTIONS', '-Xarch_host', '-fstack-protector-strong']
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, _deps(src)):
            cmd = [hipcc] + FLAGS + extra + ['-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if force or jobs or not os.path.exists(lib) or _stale(lib, [os.path.abspath(__file__)] + objs):
        # the dynamic symbol table = the functions include/ltmi.h declares: -fvisibility=hidden covers
        # the host code, the version script also hides the kernels' host-side launch stubs (hipcc
        # gives every __global__ function default visibility)
        vmap = os.path.join(objdir, 'ltmi.map')
        with open(vmap, 'w') as f:
            f.write('{\n  global:\n' + ''.join(f'    {n};\n' for n in header_exports())
                    + '  local:\n    *;\n};\n')
        run([hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', f'-Wl,--version-script={vmap}',
             '-o', lib] + extra + objs + LINK_LIBS)
    return lib


def header_path():
    """include/ltmi.h of the source tree; an installed package (no repo root above it) carries a copy
    as libertem_amd/include/ltmi.h"""
    for cand in (os.path.join(os.path.dirname(HERE), 'include', 'ltmi.h'),
                 os.path.join(HERE, 'include', 'ltmi.h')):
        if os.path.exists(cand):
            return cand
    raise RuntimeError("ltmi.h not found (looked next to the package and in libertem_amd/include)")


def header_exports():
    """names of the functions include/ltmi.h declares"""
    import re
    with open(header_path()) as f:
        hdr = f.read()
    names = set(re.findall(r'\b(ltmi_[a-z_0-9]+)\s*\(', hdr))
    return sorted(names)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, asan='--asan' in sys.argv,
                hardened='--hardened' in sys.argv))
