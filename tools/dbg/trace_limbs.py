#!/usr/bin/env python3
"""glwe_trace at many limbs: which step / Galois element leaves the oracle (bench.py --op trace --limbs 12 reported parity False with margin 0.5)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.ref import RefModule
from poulpy_amd.hal import GlweOpParams, Module
from poulpy_amd.layouts import MatZnx, VecZnx

def run(n, limbs, k, gals, batch=2):
    rng = np.random.default_rng(5)
    ref, hip = RefModule(n), Module(n, device=0)
    mat = MatZnx(n, limbs, 1, 2, limbs).fill_uniform(k, rng)
    pr, ph = ref.vmp_pmat_alloc(limbs, 1, 2, limbs), hip.vmp_pmat_alloc(limbs, 1, 2, limbs)
    ref.vmp_prepare(pr, mat); hip.vmp_prepare(ph, mat)
    a = np.stack([VecZnx(n, 2, limbs).fill_uniform(k, rng).data for _ in range(batch)])
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    p = GlweOpParams(rank=1, dnum=limbs, dsize=1, key_size=limbs, key_base2k=k, a_size=limbs, a_base2k=k, res_size=limbs, res_base2k=k, rank_out=1)
    for gl in gals:
        want = a.copy()
        for b in range(batch):
            r = VecZnx(n, 2, limbs, want[b])
            m_ref = ref.rounding_margin_of(lambda: ref.glwe_trace_assign(r, k, [gl], [pr]))
            want[b] = r.data
        d_r = hip.device_alloc(a.nbytes).upload(a)
        m = hip.rounding_margin_of(lambda: hip.glwe_trace_batched(d_r.ptr, [gl], [d_k.ptr.value], p, batch))
        hip.sync()
        got = d_r.download(np.int64, a.size).reshape(a.shape)
        print(f"n={n} limbs={limbs} base2k={k} gal={gl}: equal={np.array_equal(got, want)} gpu_margin={m:.3g} oracle_margin={m_ref:.3g} max|want|={np.abs(want).max()} max|got|={np.abs(got).max()}", flush=True)
        d_r.free()
    hip.close()

def run_full(n, limbs, k, nsteps, batch=2, calls=1):
    rng = np.random.default_rng(5)
    ref, hip = RefModule(n), Module(n, device=0)
    mat = MatZnx(n, limbs, 1, 2, limbs).fill_uniform(k, rng)
    pr, ph = ref.vmp_pmat_alloc(limbs, 1, 2, limbs), hip.vmp_pmat_alloc(limbs, 1, 2, limbs)
    ref.vmp_prepare(pr, mat); hip.vmp_prepare(ph, mat)
    a = np.stack([VecZnx(n, 2, limbs).fill_uniform(k, rng).data for _ in range(batch)])
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    p = GlweOpParams(rank=1, dnum=limbs, dsize=1, key_size=limbs, key_base2k=k, a_size=limbs, a_base2k=k, res_size=limbs, res_base2k=k, rank_out=1)
    nst = n.bit_length() - 1
    gals = ([-1] + [pow(5, 1 << i, 2 * n) for i in range(nst - 1)])[:nsteps]
    want = a.copy()
    for c in range(calls):
        for b in range(batch):
            r = VecZnx(n, 2, limbs, want[b])
            m_ref = ref.rounding_margin_of(lambda: ref.glwe_trace_assign(r, k, gals, [pr] * len(gals)))
            want[b] = r.data
    d_r = hip.device_alloc(a.nbytes).upload(a)
    for c in range(calls):
        m = hip.rounding_margin_of(lambda: hip.glwe_trace_batched(d_r.ptr, gals, [d_k.ptr.value] * len(gals), p, batch))
    hip.sync()
    got = d_r.download(np.int64, a.size).reshape(a.shape)
    print(f"FULL n={n} limbs={limbs} steps={len(gals)} calls={calls}: equal={np.array_equal(got, want)} gpu_margin={m:.3g} oracle_margin={m_ref:.3g} max|want|={np.abs(want).max()} max|got|={np.abs(got).max()}", flush=True)
    hip.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
if len(sys.argv) > 2:
    for limbs in (9, 12):
        for nsteps in (2, 4, 16):
            run_full(n, limbs, 12, nsteps)
    run_full(n, 12, 12, 16, calls=3)
else:
    for limbs in (9, 12):
        run(n, limbs, 12, [-1, 5, pow(5, 1 << 6, 2 * n), pow(5, 1 << (n.bit_length() - 3), 2 * n)])
