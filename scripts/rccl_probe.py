"""GPU box probe: dcreg_comm_init with a communicator of one rank, with and without torch in the process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "torch" in sys.argv:
    import torch
    print("torch", torch.__version__, torch.cuda.is_available())
import numpy as np
from dcreg_amd import scenes as h
from dcreg_amd import api
c = api.Context(0)
pts = h.cylinder_cloud()
c.set_target(pts, 1.0); c.set_source(pts)
uid = api.comm_unique_id()
print("uid ok", len(uid))
try:
    c.comm_init(uid, 0, 1)
    print("init ok", c.comm_allgather_sum(np.arange(32.0))[:4])
except Exception as e:
    print("FAILED:", e)
os.system("grep -i 'rccl\\|amdhip' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
