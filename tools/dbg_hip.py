import sys, os, ctypes
sys.path.insert(0, '/root/repo')
mode = sys.argv[1]
import torch
print("torch hip", torch.version.hip, "avail", torch.cuda.is_available() if mode != "noavail" else "skipped")
if mode == "tensor":
    x = torch.zeros(4, device="cuda"); print(x.sum().item())
from practical_path_guiding_lab_amd import _native as N
L = N.lib()
h = ctypes.c_void_p()
rc = L.pg_create(ctypes.byref(h), 0)
print(mode, "rc", rc, L.pg_last_error(None))
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
