"""Kernel_CUDA_Optimized on the standard benchmark inputs (main.cpp:285-352), n^3, 50 steps: prints the operator's timers.
Run under `rocprofv3 --kernel-trace` to see every launch of the time loop and the gaps between them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
zero_source = len(sys.argv) > 3 and sys.argv[3] == 'zero'
u, m, src, coords = pkg.standard_inputs(n, n, n, steps, 1)
if zero_source:
    src[:] = 0
kw = dict(dt=1e-3, h=(0.1, 0.1, 0.1), time_m=0, time_M=steps - 1)
for rep in range(3):
    u[:] = 0
    rc, timers = pkg.run_operator(L.Kernel_CUDA_Optimized, m, src, coords, u, **kw)
    assert rc == 0
    dev = timers.section0 + timers.section1
    print(f"rep {rep}: section0 {timers.section0*1e3:.3f} ms section1 {timers.section1*1e3:.3f} ms -> {n**3*(steps-5)/dev/1e9:.1f} Gcells/s")
