"""Diagnostic: a large handle, then torch's CUDA init in the same process.  python tools/probes/mem_probe.py <B>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_package()
from mpc_ilqr_mujoco_amd import solver as sv
B = int(sys.argv[1])
s = sv.BatchedILQR(B, N=25)
print("handle created, B =", B, flush=True)
import torch
print("mem_get_info", [x / 2**30 for x in torch.cuda.mem_get_info(0)], flush=True)
x = torch.zeros(8, device="cuda:0")
print("torch ok", flush=True)
s.close()
