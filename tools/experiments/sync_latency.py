import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
if len(sys.argv) > 1 and sys.argv[1] == 'spin':
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    print('setflags rc', hip.hipSetDeviceFlags(1))  # hipDeviceScheduleSpin
torch.cuda.set_device(0)
x = torch.zeros(1 << 20, device='cuda')
torch.cuda.synchronize()
for n in (1, 20):
    ts = []
    for rep in range(20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(n):
            x.add_(1.0)
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ts.append(((t1 - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
    ts.sort()
    print(n, 'kernels: wall us median %.1f, gpu events us median %.1f' % (ts[10][0], sorted(t[1] for t in ts)[10]))
