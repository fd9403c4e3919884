#!/usr/bin/env python3
"""What does one scattered 4-byte read cost on MI355X, and is the HBM fill 64 B or 128 B?
Random gathers over an 8 GiB buffer (>> L2 + Infinity Cache) through navsim_debug_gather."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch  # noqa: E402
from nav_gym_amd import lib  # noqa: E402

L = lib.load()
L.navsim_debug_gather.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
n_words = 1 << 31
x = torch.zeros(n_words, dtype=torch.float32, device="cuda:0")
n_threads = 1 << 22
iters = 16
out = torch.zeros(n_threads, dtype=torch.float32, device="cuda:0")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode, name in ((0, "1 word"), (1, "+ same 64B sector"), (2, "+ other half of the 128B line"), (3, "+ independent word")):
    ts = []
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.navsim_debug_gather(C.c_void_p(x.data_ptr()), n_words, mode, iters, n_threads, C.c_void_p(out.data_ptr()), s)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    t = min(ts[1:])
    n = n_threads * iters
    print("mode %d %-32s %.3f ms  %.2f G random reads/s  (x64B = %.2f TB/s, x128B = %.2f TB/s)"
          % (mode, name, t, n / t / 1e6, n * 64 / t / 1e9, n * 128 / t / 1e9))
