"""Per-workgroup start and end times of the last launch of a marching sweep of space order 6 .. 12
(stencil_column_march_kernel for order 6, stencil_order_march_kernel above): which workgroups a launch waits for.

    make -C accelerated-3d-acoustic-fdtd-kernel_amd clocks
    FDTD_HIP_LIB=$PWD/tools/exp/libfdtd_hip_clocks.so python tools/wg_clocks_orders.py order [n]
"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
order = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = order
dev = torch.device("cuda", 0)
npad = n + 2 * H
u_d = torch.rand((3, npad, npad, npad), device=dev) * 1e-3
m_d = torch.full((npad, npad, npad), 1.5, device=dev)
g = pkg.Geom(npad, npad, npad, 0, n - 1, 0, n - 1, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.fdtd_debug_wg_clocks_orders.argtypes = [C.c_void_p]; L.fdtd_debug_wg_clocks_orders.restype = C.c_int
out = np.zeros((3, 4096), dtype=np.uint64)
for t in range(12):
    assert L.fdtd_hip_stencil_step_order(C.byref(g), m_d.data_ptr(), u_d.data_ptr(), t % 3, (t + 2) % 3, (t + 1) % 3, order, sh) == 0
    torch.cuda.synchronize()
    if t >= 9:
        out[:] = 0
        assert L.fdtd_debug_wg_clocks_orders(out.ctypes.data) == 0
        nwg = int((out[1] > 0).sum())
        st = out[0, :nwg].astype(np.int64); en = out[1, :nwg].astype(np.int64); tile = out[2, :nwg].astype(np.int64)
        t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0
        dur = en - st
        print(f"sweep {t}: {nwg} workgroups in the last launch; start min/median/max {st.min():.1f} {np.median(st):.1f} {st.max():.1f} us; "
              f"end min/median/max {en.min():.1f} {np.median(en):.1f} {en.max():.1f} us; duration median {np.median(dur):.1f} max {dur.max():.1f}")
        print("  end percentiles 5/25/50/75/95/99:", " ".join(f"{v:.1f}" for v in np.percentile(en, [5, 25, 50, 75, 95, 99])))
        o = np.argsort(-en)[:14]
        print("  last to finish:", " ".join(f"t{tile[i]}:{st[i]:.0f}-{en[i]:.0f}" for i in o))
        o = np.argsort(en)[:8]
        print("  first to finish:", " ".join(f"t{tile[i]}:{st[i]:.0f}-{en[i]:.0f}" for i in o))
