"""Which copies of the HIP runtime and of RCCL a process ends up with, for the two import orders of torch and
libptamd.so, with a two-context group created through RCCL's loader path (dlopen by soname).
usage: python tools/runtime_probe.py torch_first | lib_first"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'cuda-pathtracer_amd', 'python'))
order = sys.argv[1] if len(sys.argv) > 1 else 'torch_first'

def mapped(tag):
    seen = sorted({l.split()[-1] for l in open('/proc/self/maps') if ('amdhip64' in l or 'rccl' in l or 'hsa-runtime' in l) and '/' in l})
    print('%-28s %s' % (tag, seen))

if order == 'torch_first':
    import torch
    mapped('after import torch')
    from ptamd import lib
    lib.lib()
    mapped('after libptamd.so')
else:
    os.environ['PTAMD_NO_TORCH_FIRST'] = '1'
    from ptamd import lib
    lib.lib()
    mapped('after libptamd.so')
    import torch
    mapped('after import torch')
print('torch sees a GPU:', torch.cuda.is_available())
import ctypes
h = ctypes.CDLL(None)
import ctypes.util
r = ctypes.CDLL('librccl.so.1', mode=ctypes.RTLD_LOCAL)
mapped('after dlopen(librccl.so.1)')
