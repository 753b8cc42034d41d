"""Diagnose HIP-runtime load/initialisation order between torch's bundled
libamdhip64 and /opt/rocm's (the one libaudioflux_mi355x.so links)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]

def maps():
    seen = set()
    for line in open("/proc/self/maps"):
        f = line.split()[-1]
        if ("amdhip64" in f or "hsa-runtime" in f) and f not in seen:
            seen.add(f)
    return sorted(seen)

import audioflux_amd as af
lib = af.get_lib()
if mode == "A":      # lib loaded, torch imported+initialised, then lib init
    import torch
    print("torch avail", torch.cuda.is_available())
    print("status", af.runtime_status())
elif mode == "B":    # lib loaded + device_count, then torch, then lib init
    print("count", lib.afx_device_count())
    import torch
    print("torch avail", torch.cuda.is_available())
    print("status", af.runtime_status())
    print(torch.zeros(4, device="cuda").sum().item())
elif mode == "C":    # lib loaded, torch imported (no init), lib init, torch init
    import torch
    print("status", af.runtime_status())
    print("torch avail", torch.cuda.is_available())
    print(torch.zeros(4, device="cuda").sum().item())
elif mode == "D":    # torch first
    pass
print(mode, maps())
