import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l})
if order == "zk_first":
    import zk_amd
    print("after zk:", maps())
    n = ctypes.c_int32(); zk_amd._lib.lib.zk_device_count(ctypes.byref(n)); print("zk count", n.value)
    try:
        c = zk_amd.Context(0, 0); print("ctx ok")
    except Exception as e:
        print("ctx fail", e, zk_amd._lib.lib.zk_last_hip_error())
    import torch
    print("after torch:", maps()); print("torch avail", torch.cuda.is_available())
else:
    import torch
    print("after torch:", maps()); print("torch avail", torch.cuda.is_available())
    import zk_amd
    print("after zk:", maps())
    n = ctypes.c_int32(); zk_amd._lib.lib.zk_device_count(ctypes.byref(n)); print("zk count", n.value)
    try:
        c = zk_amd.Context(0, 0); print("ctx ok")
    except Exception as e:
        print("ctx fail", e)
