"""Achievable HBM rates of plain streaming kernels on this GPU (torch elementwise kernels): write-only (fill), read-only (sum), copy.
   python tools/probes/hbm_bandwidth.py   -- what a kernel's WRITE_SIZE / FETCH_SIZE per time should be compared with (not the 8 TB/s headline)."""
import time
import torch

dev = torch.device("cuda:0")
n = 1 << 29                      # 4 GiB of doubles
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


gb = n * 8 / 1e9
t = timed(lambda: x.fill_(1.0));            print("write-only  fill_   %.2f GB in %.3f ms = %.2f TB/s" % (gb, t * 1e3, gb / t / 1e3))
t = timed(lambda: x.sum());                 print("read-only   sum     %.2f GB in %.3f ms = %.2f TB/s" % (gb, t * 1e3, gb / t / 1e3))
t = timed(lambda: y.copy_(x));              print("copy        copy_   %.2f GB read + %.2f GB written in %.3f ms = %.2f TB/s total" % (gb, gb, t * 1e3, 2 * gb / t / 1e3))
t = timed(lambda: torch.add(x, 1.0, out=y)); print("read+write  add     %.2f GB read + %.2f GB written in %.3f ms = %.2f TB/s total" % (gb, gb, t * 1e3, 2 * gb / t / 1e3))
