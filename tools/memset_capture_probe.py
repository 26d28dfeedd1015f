#!/usr/bin/env python3
"""Is hipMemsetAsync recorded by a stream capture on this runtime?  (docs/history.md round 6 item 8: the library clears
its buffers with kernels of its own because the answer was no.)  Prints the buffer after the capture and after each replay."""
import ctypes as C

import torch

hip = C.CDLL("libamdhip64.so")
buf = torch.full((64,), 7, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    warm = buf + 1
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st = torch.cuda.current_stream().cuda_stream
    rc = hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), C.c_int(0), C.c_size_t(256), C.c_void_p(st))
    out = buf + 1
torch.cuda.synchronize()
print("hipMemsetAsync rc", rc, "| after capture, before any replay: buf[0] =", int(buf[0]), "(7 = recorded only, 0 = executed at capture time)")
for i in range(2):
    buf.fill_(7)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print(f"replay {i}: buf[0] = {int(buf[0])}, out[0] = {int(out[0])}  (a recorded memset gives 0 and 1)")
print("torch", torch.__version__, "hip", torch.version.hip)
