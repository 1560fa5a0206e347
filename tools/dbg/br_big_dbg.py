import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from oracle.ref import RefModule
from poulpy_amd.hal import Module
from tests.test_gpu_parity import _run_blind_rotation
n = int(sys.argv[1]); k = int(sys.argv[2]); n_lwe = int(sys.argv[3]); blk = int(sys.argv[4])
ref, hip = RefModule(n), Module(n)
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 2
got, want = _run_blind_rotation(hip, ref, n, 1, n_lwe, blk, 3, 3, 3, k, batch=batch, seed=5)
d = got - want
print("n", n, "k", k, "n_lwe", n_lwe, "blk", blk, "equal", np.array_equal(got, want), "nonzero diffs", np.count_nonzero(d), "max|d|", np.abs(d).max())
if np.count_nonzero(d):
    idx = np.argwhere(d != 0)
    print("first diffs (batch, limb, col, coeff):", idx[:6].tolist(), d[d != 0][:6].tolist())
    print("diff per limb:", [int(np.count_nonzero(d[:, l])) for l in range(d.shape[1])])
