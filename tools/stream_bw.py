"""Measured HBM streaming bandwidth of the box (device-to-device copy and a read-only reduction), for the roofline section of DESIGN.md."""
import time, torch
assert torch.cuda.is_available()
n = 1 << 30                       # 4 GiB of float32 per buffer, far beyond the 256 MiB Infinity Cache
a = torch.empty(n, dtype=torch.float32, device='cuda').normal_()
b = torch.empty_like(a)
for name, fn, nbytes in (('copy (read + write)', lambda: b.copy_(a), 8*n), ('sum (read only)', lambda: a.sum(), 4*n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0)/reps
    print('%-22s %.2f TB/s' % (name, nbytes/dt/1e12))
