"""Per-workgroup start and end times of the two-step sweep's last launch, and the time every workgroup reached
each 16th plane: which workgroups a launch waits for.

    make -C accelerated-3d-acoustic-fdtd-kernel_amd clocks
    FDTD_HIP_LIB=$PWD/tools/exp/libfdtd_hip_clocks.so python tools/wg_clocks.py [n [y_m y_M [rows per workgroup]]]

Uniform tilings only (the time stamps sit in stencil_march2_kernel); one x-round of at most 1024 workgroups."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = 30
u, m, src, coords = pkg.standard_inputs(n, n, n, steps + 2, 1)
dev = torch.device("cuda", 0)
u_d = torch.from_numpy(u).to(dev); m_d = torch.from_numpy(m).to(dev)
s_d = torch.from_numpy(src).to(dev); c_d = torch.from_numpy(coords).to(dev)
ylo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
yhi = int(sys.argv[3]) if len(sys.argv) > 3 else n - 1
ntile = (yhi - ylo + 1) // (int(sys.argv[4]) if len(sys.argv) > 4 else 8)
g = pkg.Geom(n + 8, n + 8, n + 8, 0, n - 1, ylo, yhi, 0, n - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
extra = torch.zeros_like(u_d[0]); P = [u_d[0], u_d[1], u_d[2], extra]
cur, prev, third, spare = 0, 2, 1, 3
sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.fdtd_debug_wg_clocks.argtypes = [C.c_void_p]; L.fdtd_debug_wg_clocks.restype = C.c_int
out = np.zeros((14, 1024), dtype=np.uint64)
for t in range(0, steps, 2):
    rc = L.fdtd_hip_two_steps_ex(C.byref(g), m_d.data_ptr(), s_d.data_ptr(), c_d.data_ptr(), P[cur].data_ptr(), P[prev].data_ptr(),
                                 P[third].data_ptr(), P[spare].data_ptr(), t, 0, 0, 1, 3, -1, 1, None, None, sh)
    assert rc == 0
    cur, prev, third, spare = spare, third, cur, prev
    torch.cuda.synchronize()
    if t >= steps - 6:
        assert L.fdtd_debug_wg_clocks(out.ctypes.data) == 0
        nwg = int((out[1] > 0).sum())
        st = out[0, :nwg].astype(np.int64); en = out[1, :nwg].astype(np.int64)
        xcc = (out[2, :nwg] >> np.uint64(32)).astype(np.int64) & 0xf; tile = (out[2, :nwg] & np.uint64(0xffffffff)).astype(np.int64)
        t0 = st.min()
        st = (st - t0) / 100.0; en = (en - t0) / 100.0     # us (100 MHz)
        print(f"pair at t={t}: {nwg} workgroups; start spread {st.max():.1f} us; end min/median/max {en.min():.1f} {np.median(en):.1f} {en.max():.1f} us")
        q = np.percentile(en, [5, 25, 50, 75, 95])
        print("  end percentiles 5/25/50/75/95:", " ".join(f"{v:.1f}" for v in q))
        order = np.argsort(-en)[:12]
        print("  slowest:", " ".join(f"t{tile[i]}(x{xcc[i]}):{en[i]:.0f}" for i in order))
        order = np.argsort(en)[:6]
        print("  fastest:", " ".join(f"t{tile[i]}(x{xcc[i]}):{en[i]:.0f}" for i in order))
        ph = (out[3:11, :nwg].astype(np.int64) - t0) / 100.0
        for name, sel in (("tile0", (tile % ntile) == 0), ("tile63", (tile % ntile) == ntile - 1), ("others", ((tile % ntile) > 4) & ((tile % ntile) < ntile - 5))):
            print("  ", name, "time at the end of each 16-plane bucket:", " ".join(f"{np.median(ph[b][sel]):.0f}" for b in range(8)))
        ph2 = (out[11:14, :nwg].astype(np.int64) - t0) / 100.0
        print(f"   medians: prologue done {np.median(ph2[0] - st):.1f} us after start, lead-in (4 planes) {np.median(ph2[1] - ph2[0]):.1f} us, "
              f"steady planes {np.median(ph2[2] - ph2[1]):.1f} us, tail planes {np.median(en - ph2[2]):.1f} us")
