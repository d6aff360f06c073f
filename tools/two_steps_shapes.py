import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import torch
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
reps = 60
for shape in sys.argv[1:]:
    nx, ny, nz = (int(v) for v in shape.split("x"))
    ps = (nx + 8, ny + 8, nz + 8)
    lv = [torch.rand(ps, device="cuda") * 0.01 for _ in range(4)]
    m = torch.rand(ps, device="cuda") + 1
    g = pkg.Geom(*ps, 0, nx - 1, 0, ny - 1, 0, nz - 1, 1e-3, 0.1, 0.1, 0.1, 0.0, 0.0, 0.0)
    out = []
    for var in ("march", "march2"):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for r in range(reps + 5):
            if r == 5: e0.record()
            a, b, c, d = (lv[(r * 2 + i) % 4] for i in range(4))   # rotate the buffers like a time loop
            rc = L.fdtd_hip_two_steps(C.byref(g), m.data_ptr(), None, None, a.data_ptr(), b.data_ptr(), c.data_ptr(),
                                      d.data_ptr(), 0, 0, -1, 1, 3, pkg.VARIANTS[var], None)
            assert rc == 0
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out.append(f"{var} {ms:.4f} ms {2*nx*ny*nz/ms/1e6:.1f} Gcell/s")
    print(f"SHAPE {shape}: " + " | ".join(out), flush=True)
