#!/usr/bin/env python3
"""mi_ode_outer_reduce alone at config 4's shape (65536 x 128, float64 / float32): time per call by HIP events, against torch."""
import torch

from tfdiffeq_amd import _native as N

lib = N.load()
dev = torch.device('cuda:0')
for dtype in (torch.float64, torch.float32):
    B, D = 65536, 128
    y = torch.randn(B, D, dtype=dtype, device=dev)
    a = torch.randn(B, D, dtype=dtype, device=dev)
    code = N.dtype_code(dtype)
    ws = torch.empty(int(lib.mi_ode_outer_workspace_bytes(code, B, D)), dtype=torch.uint8, device=dev)
    w_out = torch.empty(D, D, dtype=dtype, device=dev)
    b_out = torch.empty(D, dtype=dtype, device=dev)

    def ours():
        N.check(lib.mi_ode_outer_reduce(code, B, D, y.data_ptr(), a.data_ptr(), -1.0, w_out.data_ptr(), b_out.data_ptr(), ws.data_ptr(),
                                        N.stream_ptr(dev)), 'mi_ode_outer_reduce')

    def blas():
        return y.t() @ a, a.sum(0)
    for name, fn in (('mi_ode_outer_reduce', ours), ('torch (rocBLAS)', blas)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%s %-20s %8.1f us per call   (%.2f TB/s of the two planes, %.1f TFLOP/s)' % (
            str(dtype)[6:], name, us, 2 * B * D * y.element_size() / us / 1e6, 2.0 * B * D * D / us / 1e6))
