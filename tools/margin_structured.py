#!/usr/bin/env python3
"""Rounding margin and parity on STRUCTURED inputs (tools/structured.py: all-minimum digits, alternating signs, single tones, a delta), for the
BASELINE shapes: GPU margin (pz_module_set_margin_probe), the ORACLE's own margin (pzr_margin_probe_*), bit parity GPU == oracle, and - for
the GLWE products - oracle == the exact integer product (no FFT).  Sweeps base2k upwards per shape until a structured input breaks parity or
its margin reaches 0.25.  VERDICT r05 item 3; poulpy-hal/docs/backend_safety_contract.md:25-27 ("fp tolerance must be documented").

    python tools/margin_structured.py [--shapes metric,config1,br_ref,br_big] [--out gpurun_out/margin_structured.md] [--quick]

Every polynomial of the ciphertext AND of the key carries the same pattern (the coherent worst case).  Blind rotations: key digits and test
vector follow the pattern, the LWE exponents are random (they only select monomials)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import structured as st   # noqa: E402

UNSAFE = 0.25


def glwe_case(hip, ref, n, rank, size, base2k, name, exact=True):
    """One external product (dnum = size, dsize 1) with pattern `name` in every polynomial of ciphertext and key."""
    from poulpy_amd.hal import GlweOpParams
    from poulpy_amd.layouts import MatZnx, VecZnx
    cols = rank + 1
    mat = MatZnx(n, size, cols, cols, size)
    a = VecZnx(n, cols, size)
    if name == "uniform":
        rng = np.random.default_rng(7 * base2k + n)
        mat.fill_uniform(base2k, rng)
        a.fill_uniform(base2k, rng)
    else:
        mat.data[...] = st.fill(mat.data.shape[:-1], name, n, base2k)
        a.data[...] = st.fill(a.data.shape[:-1], name, n, base2k)
    pr, ph = ref.vmp_pmat_alloc(size, cols, cols, size), hip.vmp_pmat_alloc(size, cols, cols, size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    want = VecZnx(n, cols, size)
    m_ref = ref.rounding_margin_of(lambda: ref.glwe_external_product(want, base2k, a, base2k, pr, 1, base2k))
    batch = 2
    a_all = np.stack([a.data] * batch)
    d_a = hip.device_alloc(a_all.nbytes).upload(a_all)
    d_k = hip.device_alloc(ph.data.nbytes).upload(ph.data)
    d_r = hip.device_alloc(a_all.nbytes)
    p = GlweOpParams(rank=rank, dnum=size, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=rank)
    try:
        m_gpu = hip.rounding_margin_of(lambda: hip.glwe_external_product_batched(d_r.ptr, d_a.ptr, d_k.ptr, p, batch))
        hip.sync()
        got = d_r.download(np.int64, a_all.size).reshape(a_all.shape)
    finally:
        for buf in (d_a, d_k, d_r):
            buf.free()
    out = {"gpu_margin": m_gpu, "oracle_margin": m_ref, "gpu_eq_oracle": bool(np.array_equal(got[0], want.data) and np.array_equal(got[1], want.data))}
    if exact:
        ex = st.exact_external_product(a.data, mat.data, base2k)
        out["oracle_eq_exact"] = bool(np.array_equal(ex, want.data))
        out["gpu_eq_exact"] = bool(np.array_equal(ex, got[0]))
    return out


def br_case(hip, ref, n, rank, block, dnum, brk_size, res_size, base2k, name, n_lwe):
    """CGGI block-binary blind rotation with pattern `name` in every key digit polynomial and in the test vector."""
    from poulpy_amd.hal import BlindRotationParams
    from poulpy_amd.layouts import MatZnx, VecZnx
    cols = rank + 1
    rng = np.random.default_rng(11 * base2k + n)
    pm_len = n * dnum * cols * cols * brk_size
    brk_ref = np.empty((n_lwe, pm_len), dtype=np.float64)
    brk_hip = np.empty((n_lwe, pm_len), dtype=np.float64)
    mat = MatZnx(n, dnum, cols, cols, brk_size)
    if name == "uniform":
        mat.fill_uniform(base2k, rng)
    else:
        mat.data[...] = st.fill(mat.data.shape[:-1], name, n, base2k)
    pr, ph = ref.vmp_pmat_alloc(dnum, cols, cols, brk_size), hip.vmp_pmat_alloc(dnum, cols, cols, brk_size)
    ref.vmp_prepare(pr, mat)
    hip.vmp_prepare(ph, mat)
    brk_ref[:] = pr.data.reshape(-1)      # (every GGSW of the key alike: the coherent worst case)
    brk_hip[:] = ph.data.reshape(-1)
    lut = VecZnx(n, 1, res_size)
    if name == "uniform":
        lut.fill_uniform(base2k, rng)
    else:
        lut.data[...] = st.fill(lut.data.shape[:-1], name, n, base2k)
    batch = 2
    lwe = rng.integers(-n, n, (batch, n_lwe + 1), dtype=np.int64)   # mod_switch_2n output range
    want = np.empty((batch, res_size, cols, n), dtype=np.int64)
    x_pow_a = ref.blind_rotation_x_pow_a()
    m_ref = 0.0
    for b in range(batch):
        r = VecZnx(n, cols, res_size)
        m_ref = max(m_ref, ref.rounding_margin_of(lambda: ref.blind_rotation_execute(r, base2k, np.ascontiguousarray(lwe[b]), lut, brk_ref, dnum, brk_size, block, x_pow_a)))
        want[b] = r.data
    d_lwe = hip.device_alloc(lwe.nbytes).upload(lwe)
    d_lut = hip.device_alloc(lut.data.nbytes).upload(lut.data)
    d_brk = hip.device_alloc(brk_hip.nbytes).upload(brk_hip)
    d_res = hip.device_alloc(want.nbytes)
    p = BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block, dnum=dnum, brk_size=brk_size, base2k=base2k, res_size=res_size, lut_size=res_size)
    try:
        m_gpu = hip.rounding_margin_of(lambda: hip.blind_rotation_execute_batched(d_res.ptr, d_lwe.ptr, d_lut.ptr, d_brk.ptr, p, batch))
        hip.sync()
        got = d_res.download(np.int64, want.size).reshape(want.shape)
    finally:
        for buf in (d_lwe, d_lut, d_brk, d_res):
            buf.free()
    return {"gpu_margin": m_gpu, "oracle_margin": m_ref, "gpu_eq_oracle": bool(np.array_equal(got, want))}


SHAPES = {
    # label, runner(hip, ref, k, name), N, the BASELINE base2k value(s), sweep
    "metric": ("external product, N=2^16, 8 limbs, rank 1 (metric)", lambda h, r, k, nm: glwe_case(h, r, 65536, 1, 8, k, nm), 65536, (12, 14), range(12, 20)),
    "config1": ("external product, N=2^12, 4 limbs, rank 1 (configs[1])", lambda h, r, k, nm: glwe_case(h, r, 4096, 1, 4, k, nm), 4096, (17,), range(14, 23)),
    "br_ref": ("blind rotation `ref`: N=512, rank 3, block 3, key 2 limbs (configs[3], reference bench)", lambda h, r, k, nm: br_case(h, r, 512, 3, 3, 1, 2, 1, k, nm, 9), 512, (18,), range(14, 24)),
    "br_big": ("blind rotation `big`: N=2^14, rank 1, block 7, dnum 3, 3 limbs (configs[3])", lambda h, r, k, nm: br_case(h, r, 16384, 1, 7, 3, 3, 3, k, nm, 14), 16384, (13,), range(12, 20)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="metric,config1,br_ref,br_big")
    ap.add_argument("--out", default="gpurun_out/margin_structured.md")
    ap.add_argument("--quick", action="store_true", help="the BASELINE base2k values only")
    args = ap.parse_args()
    from oracle.ref import RefModule
    from poulpy_amd.hal import Module
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    jl = open(os.path.splitext(args.out)[0] + ".jsonl", "w")
    lines = ["| shape | base2k | input | GPU margin | oracle margin | GPU = oracle | oracle = exact | GPU = exact |", "|---|---|---|---|---|---|---|---|"]
    summary = []
    for key in args.shapes.split(","):
        label, run, n, base_ks, sweep = SHAPES[key]
        hip, ref = Module(n, device=0), RefModule(n)
        first_break, first_unsafe, worst_at_base = None, None, {}
        ks = list(base_ks) if args.quick else sorted(set(base_ks) | set(sweep))
        for k in ks:
            stop = False
            for name in ("uniform",) + st.PATTERNS:
                t0 = time.time()
                r = run(hip, ref, k, name)
                r.update(shape=key, base2k=k, input=name, seconds=round(time.time() - t0, 2))
                jl.write(json.dumps(r) + "\n")
                jl.flush()
                print(json.dumps(r), flush=True)
                lines.append(f"| {label} | {k} | {name} | {r['gpu_margin']:.2e} | {r['oracle_margin']:.2e} | {r['gpu_eq_oracle']} | {r.get('oracle_eq_exact', '-')} | {r.get('gpu_eq_exact', '-')} |")
                if name != "uniform":
                    w = max(r["gpu_margin"], r["oracle_margin"])
                    if k in base_ks:
                        prev = worst_at_base.get(k, (0.0, ""))
                        if w >= prev[0]:
                            worst_at_base[k] = (w, name)
                    if first_unsafe is None and w >= UNSAFE:
                        first_unsafe = (k, name)
                    broke = (not r["gpu_eq_oracle"]) or (r.get("oracle_eq_exact") is False) or (r.get("gpu_eq_exact") is False)
                    if first_break is None and broke:
                        first_break = (k, name, {x: r.get(x) for x in ("gpu_eq_oracle", "oracle_eq_exact", "gpu_eq_exact")})
                        stop = True
            if stop and k not in base_ks and k > max(base_ks):
                break
        summary.append({"shape": key, "label": label, "worst_structured_margin_at_baseline_base2k": {str(k): v for k, v in worst_at_base.items()},
                        "first_unsafe_structured": first_unsafe, "first_break": first_break})
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n\n")
        f.write("| shape | worst structured margin at the BASELINE base2k (input) | first base2k with a structured margin >= 0.25 | first base2k where a structured input breaks parity / exactness |\n|---|---|---|---|\n")
        for s in summary:
            wb = "; ".join(f"base2k {k}: {v[0]:.2e} ({v[1]})" for k, v in s["worst_structured_margin_at_baseline_base2k"].items())
            f.write(f"| {s['label']} | {wb} | {s['first_unsafe_structured']} | {s['first_break']} |\n")
    jl.write(json.dumps({"summary": summary}) + "\n")
    print(open(args.out).read())


if __name__ == "__main__":
    main()
