import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
first = sys.argv[1]
if first == "torch":
    import torch
from ygz_slam_amd import _lib
ctx = _lib.HipContext(width=320, height=240, levels=3, max_frames=2)
ctx.close()
import torch
try:
    print(first, "first:", torch.zeros(2, device="cuda").sum().item(), "ok")
except Exception as e:
    print(first, "first: FAILED", e)
os.system("grep -c amdhip64 /proc/%d/maps; grep amdhip64 /proc/%d/maps | awk '{print $6}' | sort -u" % (os.getpid(), os.getpid()))
