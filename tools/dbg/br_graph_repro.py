"""Round 6: the structured blind-rotation cases at N = 2^14 (batch 2, two blocks) fail after other work in the same process, only with HIP-graph
replay on (tools/dbg/r6_run7.sh: round-5 library alike).  This script replays the pytest order and prints, per case, the device pointers, the graph
launches so far and the verdict."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import margin_structured as ms
import structured as st
from oracle.ref import RefModule
from poulpy_amd.hal import Module, DeviceBuffer

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if "torch" in sys.argv[2:]:   # as tests/conftest.py: PyTorch's own HIP runtime (7.0) initialises first and serves libpoulpy_hip.so too
    import torch
    torch.cuda.init()
    print("torch", torch.__version__, "hip", torch.version.hip, flush=True)
_orig_init = DeviceBuffer.__init__
def _init(self, module, nbytes):
    _orig_init(self, module, nbytes)
    print("      alloc %10d B at %#x" % (nbytes, self.ptr.value), flush=True)
DeviceBuffer.__init__ = _init

mods = {}
def get(n):
    if n not in mods:
        mods[n] = (Module(n), RefModule(n))
    return mods[n]

if mode != "br_only":
    for (n, size, k) in ((65536, 8, 12), (65536, 8, 14), (4096, 4, 17)):
        hip, ref = get(n)
        for name in st.PATTERNS:
            r = ms.glwe_case(hip, ref, n, 1, size, k, name, exact=False)
            print("ext", n, name, r["gpu_eq_oracle"], flush=True)
    hip, ref = get(512)
    for name in st.PATTERNS:
        r = ms.SHAPES["br_ref"][1](hip, ref, 18, name)
        print("br_ref", name, r["gpu_eq_oracle"], flush=True)
hip, ref = get(16384)
if mode == "nographs":
    hip.lib.pz_module_set_graphs(hip.handle, 0)
for rep in range(2):
    for name in st.PATTERNS:
        g0 = hip.graph_launches()
        r = ms.SHAPES["br_big"][1](hip, ref, 13, name)
        print("br_big", rep, name, "graph launches %d -> %d" % (g0, hip.graph_launches()), r, flush=True)
