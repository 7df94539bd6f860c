"""Registers, scratch and LDS of the kernels in libltmi.so whose (mangled) name contains a pattern:
    python scripts/kernel_resources.py k_dense_fold [path/to/lib.so]"""
import os, re, subprocess, sys, tempfile
pat = sys.argv[1] if len(sys.argv) > 1 else ''
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                         'libertem_amd', '_lib', 'libltmi.so')
llvm = '/opt/rocm/lib/llvm/bin/'
notes = ''
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, 'fat.bin')
    subprocess.check_call([llvm + 'llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, lib, os.path.join(d, 'x')])
    blob = open(fat, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    starts = [m.start() for m in re.finditer(magic, blob)]          # one bundle per translation unit
    for i, st in enumerate(starts):
        part, co = os.path.join(d, f'b{i}.bin'), os.path.join(d, f'b{i}.co')
        open(part, 'wb').write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.check_call([llvm + 'clang-offload-bundler', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                               '--input=' + part, '--output=' + co, '--unbundle'])
        notes += subprocess.check_output([llvm + 'llvm-readelf', '--notes', co], text=True)
for blk in notes.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk)
    if not name or pat not in name.group(1):
        continue
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', blk).group(1)
    dem = subprocess.run(['c++filt', name.group(1)], capture_output=True, text=True).stdout.strip()
    agpr = re.match(r':\s+(\d+)', blk).group(1)
    print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>3s} agpr {agpr:>3s} sgpr {g('sgpr_count'):>3s} "
          f"spill {g('vgpr_spill_count')} scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
