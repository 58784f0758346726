"""The transform pair of channel_conv_kernel alone (forward 8192-point transform, spectrum pass, inverse) on every CU: clocks per phase
and pairs per second.  Needs the instrumented library:   NRHIP_LIB_NAME=libnrhip_ct.so ./build.sh -DNRHIP_CONV_TIMING
    python tools/conv_pair_probe.py [n_iter] [L]"""
import ctypes
import os
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
os.environ['NRHIP_LIB_NAME'] = os.environ.get('NRHIP_LIB_NAME', 'libnrhip_ct.so')
import nuradiomc_amd  # noqa: E402
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5296
ctx = nuradiomc_amd.Context((1.78, 0.423, 77.), 'SP1')
h = ctypes.CDLL(os.path.join(ROOT, 'nuradiomc_amd', 'lib', os.environ['NRHIP_LIB_NAME']))
h.nrhip_debug_conv_pair.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_float)]
for variant in (0, 1):
    clk = (ctypes.c_ulonglong * 6)()
    ms = ctypes.c_float()
    assert h.nrhip_debug_conv_pair(ctx._h, n_iter, L, variant, clk, ctypes.byref(ms)) == 0
    nb = 256
    per = [c / (nb * n_iter) for c in clk]
    print('variant %d (%s): %.3f ms for %d pairs on %d blocks = %.2f us per pair per CU; clocks per pair: wave 0 fwd %.0f mid %.0f inv %.0f | wave 5 fwd %.0f mid %.0f inv %.0f (sum %.0f)'
          % (variant, 'one response table, L2-hot' if variant == 0 else '64 tables in turn', ms.value, nb * n_iter, nb, 1e3 * ms.value / n_iter,
             per[0], per[1], per[2], per[3], per[4], per[5], sum(per[:3])))
