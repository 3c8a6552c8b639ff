import ctypes, os, sys
sys.path.insert(0, '/root/repo')
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l))
mode = sys.argv[1]
if mode == 'lib_only':
    L = ctypes.CDLL('/root/repo/witch_amd/libwitch_hip.so')
    print('maps', maps())
    print('wh_init', L.wh_init(0))
    L.wh_last_error.restype = ctypes.c_char_p
    print(L.wh_last_error())
else:
    import torch
    print('torch', torch.__version__, torch.cuda.is_available(), torch.cuda.device_count())
    print('maps after torch', maps())
    L = ctypes.CDLL('/root/repo/witch_amd/libwitch_hip.so')
    print('maps after lib', maps())
    print('wh_init', L.wh_init(0))
    L.wh_last_error.restype = ctypes.c_char_p
    print(L.wh_last_error())
