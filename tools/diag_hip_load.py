import sys, os, ctypes
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l})
if order == "gsf_first":
    from gps_optimize_slam_amd import _lib
    L = _lib.load(); print("after load:", maps())
    import torch
    print("torch avail", torch.cuda.is_available(), maps())
    print("gsf count", L.gsf_device_count())
elif order == "gsf_first_init":
    from gps_optimize_slam_amd import _lib
    L = _lib.load(); print("gsf count", L.gsf_device_count(), maps())
    import torch
    print("torch avail", torch.cuda.is_available(), maps())
else:
    import torch
    print("torch avail", torch.cuda.is_available(), maps())
    from gps_optimize_slam_amd import _lib
    L = _lib.load(); print("gsf count", L.gsf_device_count(), maps())
