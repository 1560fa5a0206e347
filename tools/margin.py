#!/usr/bin/env python3
"""Rounding margin of every op family of the path, on the kernels the product path dispatches: max |x - round(x)| over the values the
inverse transforms round (pz_module_set_margin_probe), for a sweep of base2k per shape.  0.5 would be a wrong i64 limb; the FFT64
family is exact only while this stays well below it ("fp tolerance must be documented", poulpy-hal/docs/backend_safety_contract.md:25-27;
SURVEY.md 7 "exactness margin").  The error grows like N * rows * 2^(2 base2k): about x4 per extra bit of base2k.

    python tools/margin.py [--families glwe,auto,tensor,br,hal] [--out gpurun_out/margin_table.md] [--quick]

Inputs are uniform digits in [-2^(base2k-1), 2^(base2k-1)) (fill_uniform, the distribution of the reference's own tests); `unsafe from` =
the first base2k of the sweep whose margin reaches 0.25 (one more bit would pass 0.5).  A few ciphertexts per shape: the margin is a
maximum over N * limbs * columns * batch values per call and moves little with the batch."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

UNSAFE = 0.25
P = C.c_void_p


def ptr(t):
    return P(t.data_ptr())


class Ctx:
    def __init__(self):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", 0)
        self.mods = {}

    def mod(self, n):
        from poulpy_amd.hal import Module
        if n not in self.mods:
            self.mods[n] = Module(n, device=0)
        return self.mods[n]

    def digits(self, shape, base2k, seed):
        g = self.torch.Generator(device=self.dev)
        g.manual_seed(seed)
        half = 1 << (base2k - 1)
        return self.torch.randint(-half, half, shape, dtype=self.torch.int64, device=self.dev, generator=g)

    def prepare(self, mod, n, rows, cols_in, cols_out, size, base2k, seed):
        mat = self.digits((n * rows * cols_in * cols_out * size,), base2k, seed)
        pmat = self.torch.empty(mat.numel(), dtype=self.torch.float64, device=self.dev)
        self.torch.cuda.synchronize()
        mod._ck(mod.lib.pz_vmp_prepare(mod.handle, ptr(pmat), ptr(mat), C.c_size_t(rows), C.c_size_t(cols_in), C.c_size_t(cols_out), C.c_size_t(size)))
        mod.sync()
        return pmat


def glwe_margin(cx, op, n, rank, size, base2k, batch, gal=5):
    """external_product | keyswitch | automorphism | automorphism_add at (n, rank, size limbs, dnum = size, dsize 1)."""
    from poulpy_amd.hal import GlweOpParams
    mod = cx.mod(n)
    cols = rank + 1
    cols_in = cols if op == "external_product" else rank
    key = cx.prepare(mod, n, size, cols_in, cols, size, base2k, 11 * base2k + size)
    a = cx.digits((batch, size, cols, n), base2k, 13 * base2k + size)
    res = cx.torch.empty_like(a)
    cx.torch.cuda.synchronize()
    p = GlweOpParams(rank=rank, dnum=size, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                     res_base2k=base2k, rank_out=rank)
    if op == "external_product":
        run = lambda: mod.glwe_external_product_batched(ptr(res), ptr(a), ptr(key), p, batch)
    elif op == "keyswitch":
        run = lambda: mod.glwe_keyswitch_batched(ptr(res), ptr(a), ptr(key), p, batch)
    else:
        mode = "add" if op == "automorphism_add" else "automorphism"
        run = lambda: mod.glwe_automorphism_batched(ptr(res), ptr(a), ptr(key), p, gal, mode, batch)
    mod.dispatch_notes(reset=True)
    m = mod.rounding_margin_of(run)
    return m, mod.dispatch_notes()


def tensor_margin(cx, n, size, base2k, batch, relin):
    from poulpy_amd.hal import GlweOpParams, GlweTensorParams
    mod = cx.mod(n)
    rank, cols, tcols = 1, 2, 3
    a = cx.digits((batch, size, cols, n), base2k, 17 * base2k + size)
    b = cx.digits((batch, size, cols, n), base2k, 19 * base2k + size)
    res = cx.torch.zeros((batch, size, tcols, n), dtype=cx.torch.int64, device=cx.dev)
    p = GlweTensorParams(rank=rank, a_size=size, b_size=size, ab_base2k=base2k, a_effective_k=size * base2k, b_effective_k=size * base2k,
                         res_size=size, res_base2k=base2k, cnv_offset=size * base2k - 20)
    key = out = rp = None
    if relin:
        key = cx.prepare(mod, n, size, 1, cols, size, base2k, 23 * base2k + size)
        out = cx.torch.zeros((batch, size, cols, n), dtype=cx.torch.int64, device=cx.dev)
        rp = GlweOpParams(rank=rank, dnum=size, dsize=1, key_size=size, key_base2k=base2k, a_size=size, a_base2k=base2k, res_size=size,
                          res_base2k=base2k, rank_out=rank)
    cx.torch.cuda.synchronize()

    def run():
        mod.glwe_tensor_apply_batched(ptr(res), ptr(a), ptr(b), p, "apply", batch)
        if relin:
            mod.glwe_tensor_relinearize_batched(ptr(out), ptr(res), ptr(key), rp, batch)
    mod.dispatch_notes(reset=True)
    m = mod.rounding_margin_of(run)
    return m, mod.dispatch_notes()


def br_margin(cx, n, rank, block_size, dnum, brk_size, res_size, base2k, batch, n_lwe=None):
    """CGGI blind rotation on a short LWE (three blocks: every block step rounds and normalizes on its own, the margin does not build up
    over the blocks), the key and table digits uniform."""
    from poulpy_amd.hal import BlindRotationParams
    mod = cx.mod(n)
    cols = rank + 1
    n_lwe = n_lwe or 3 * block_size
    pm = n * dnum * cols * cols * brk_size
    brk = cx.torch.empty((n_lwe, pm), dtype=cx.torch.float64, device=cx.dev)
    for i in range(n_lwe):
        mat = cx.digits((pm,), base2k, 29 * base2k + i)
        cx.torch.cuda.synchronize()
        mod._ck(mod.lib.pz_vmp_prepare(mod.handle, ptr(brk[i]), ptr(mat), C.c_size_t(dnum), C.c_size_t(cols), C.c_size_t(cols), C.c_size_t(brk_size)))
        mod.sync()
    lut = cx.digits((res_size, 1, n), base2k, 31 * base2k)
    g = cx.torch.Generator(device=cx.dev)
    g.manual_seed(37)
    lwe = cx.torch.randint(-n, n, (batch, n_lwe + 1), dtype=cx.torch.int64, device=cx.dev, generator=g)
    res = cx.torch.empty((batch, res_size, cols, n), dtype=cx.torch.int64, device=cx.dev)
    p = BlindRotationParams(rank=rank, n_lwe=n_lwe, block_size=block_size, dnum=dnum, brk_size=brk_size, base2k=base2k, res_size=res_size,
                            lut_size=res_size)
    cx.torch.cuda.synchronize()
    mod.dispatch_notes(reset=True)
    m = mod.rounding_margin_of(lambda: mod.blind_rotation_execute_batched(ptr(res), ptr(lwe), ptr(lut), ptr(brk), p, batch))
    return m, mod.dispatch_notes()


def hal_idft_margin(cx, n, size, base2k, batch):
    """per-op path: dft_apply x 2 columns, vmp_apply_dft_to_dft (rows = size), idft_apply_consume - the reference's own op sequence."""
    mod = cx.mod(n)
    cols = 2
    key = cx.prepare(mod, n, size, cols, cols, size, base2k, 41 * base2k + size)
    a = cx.digits((batch, size, cols, n), base2k, 43 * base2k + size)
    a_dft = cx.torch.empty((batch, size, cols, n), dtype=cx.torch.float64, device=cx.dev)
    r_dft = cx.torch.empty((batch, size, cols, n), dtype=cx.torch.float64, device=cx.dev)
    cx.torch.cuda.synchronize()

    def run():
        for c in range(cols):
            mod.vec_znx_dft_apply_batched(batch, 1, 0, ptr(a_dft), cols, size, c, ptr(a), cols, size, c)
        mod.vmp_apply_dft_to_dft_batched(batch, ptr(r_dft), cols, size, ptr(a_dft), cols, size, ptr(key), size, cols, cols, size, 0)
        mod.vec_znx_idft_apply_consume_batched(batch, ptr(r_dft), cols, size)
    mod.dispatch_notes(reset=True)
    m = mod.rounding_margin_of(run)
    return m, mod.dispatch_notes()


def kernels_of(notes):
    """the kernel names of a dispatch-note string, shortened"""
    names = []
    for part in (notes or "").split(";"):
        part = part.strip()
        if part:
            nm = part.split(" ")[0]
            if nm not in names:
                names.append(nm)
    return ", ".join(names)[:120]


def sweep(label, fn, base2ks, rows, log):
    first_unsafe = None
    cells = []
    notes = ""
    for k in base2ks:
        try:
            m, notes_k = fn(k)
        except Exception as e:   # a shape the library refuses at this base2k (e.g. 32-bit accumulators beyond base2k 29): end of the sweep
            cells.append((k, None, str(e)[:60]))
            break
        notes = notes or notes_k
        cells.append((k, m, None))
        log({"shape": label, "base2k": k, "margin": m, "dispatch": notes_k})
        if first_unsafe is None and m >= UNSAFE:
            first_unsafe = k
            break
    rows.append((label, cells, first_unsafe, notes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", default="glwe,auto,tensor,br,hal")
    ap.add_argument("--out", default="gpurun_out/margin_table.md")
    ap.add_argument("--quick", action="store_true", help="two base2k per shape (smoke run)")
    args = ap.parse_args()
    fam = set(args.families.split(","))
    cx = Ctx()
    rows = []
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    jl = open(os.path.splitext(args.out)[0] + ".jsonl", "w")

    def log(d):
        jl.write(json.dumps(d) + "\n")
        jl.flush()
        print(json.dumps({k: v for k, v in d.items() if k != "dispatch"}), flush=True)
    q = (lambda ks: ks[:2]) if args.quick else (lambda ks: ks)
    r12_22 = list(range(12, 23))
    if "glwe" in fam:
        sweep("external product, N=2^16, 8 limbs, rank 1 (metric; configs[2] key switch below)", lambda k: glwe_margin(cx, "external_product", 65536, 1, 8, k, 8), q(r12_22), rows, log)
        sweep("key switch, N=2^16, 8 limbs, rank 1 (configs[2])", lambda k: glwe_margin(cx, "keyswitch", 65536, 1, 8, k, 8), q(r12_22), rows, log)
        sweep("external product, N=2^16, 16 limbs, rank 1", lambda k: glwe_margin(cx, "external_product", 65536, 1, 16, k, 4), q(r12_22), rows, log)
        sweep("external product, N=2^12, 4 limbs, rank 1 (configs[1], base2k 17)", lambda k: glwe_margin(cx, "external_product", 4096, 1, 4, k, 32), q(list(range(14, 27))), rows, log)
        sweep("external product, N=2^11, 3 limbs, rank 1 (small-ring pipeline)", lambda k: glwe_margin(cx, "external_product", 2048, 1, 3, k, 32), q(list(range(14, 27))), rows, log)
        sweep("external product, N=2^10, 2 limbs, rank 1 (configs[0] shape)", lambda k: glwe_margin(cx, "external_product", 1024, 1, 2, k, 32), q(list(range(16, 28))), rows, log)
        sweep("external product, N=2^13, 4 limbs, rank 1", lambda k: glwe_margin(cx, "external_product", 8192, 1, 4, k, 16), q(list(range(14, 26))), rows, log)
    if "auto" in fam:
        sweep("glwe_automorphism (p=5), N=2^16, 16 limbs (configs[4] rotate)", lambda k: glwe_margin(cx, "automorphism", 65536, 1, 16, k, 4), q(r12_22), rows, log)
        sweep("glwe_automorphism_add (p=-1), N=2^16, 8 limbs (trace step)", lambda k: glwe_margin(cx, "automorphism_add", 65536, 1, 8, k, 8, gal=-1), q(r12_22), rows, log)
        sweep("glwe_automorphism (p=5), N=2^10, 3 limbs (circuit bootstrapping trace)", lambda k: glwe_margin(cx, "automorphism", 1024, 1, 3, k, 32), q(list(range(13, 27))), rows, log)
    if "tensor" in fam:
        sweep("glwe_tensor_apply, N=2^16, 16 limbs (configs[4] multiply, tensoring)", lambda k: tensor_margin(cx, 65536, 16, k, 3, False), q(r12_22), rows, log)
        sweep("glwe_tensor_apply + relinearize, N=2^16, 16 limbs (configs[4] multiply)", lambda k: tensor_margin(cx, 65536, 16, k, 3, True), q(r12_22), rows, log)
        sweep("glwe_tensor_apply, N=2^12, 4 limbs", lambda k: tensor_margin(cx, 4096, 4, k, 8, False), q(list(range(14, 27))), rows, log)
    if "br" in fam:
        k18 = list(range(13, 28))
        sweep("blind rotation `ref`: N=512, rank 3, block 3, dnum 1, key 2 limbs (reference bench, base2k 18)", lambda k: br_margin(cx, 512, 3, 3, 1, 2, 1, k, 64), q(k18), rows, log)
        sweep("blind rotation `cbt`: N=1024, rank 1, block 7, dnum 3, 3 limbs (one-kernel path)", lambda k: br_margin(cx, 1024, 1, 7, 3, 3, 3, k, 64), q(k18), rows, log)
        sweep("blind rotation: N=1024, rank 2, block 7, dnum 3, 4 limbs (circuit-bootstrapping key)", lambda k: br_margin(cx, 1024, 2, 7, 3, 4, 4, k, 64), q(k18), rows, log)
        sweep("blind rotation: N=2048, rank 1, block 7, dnum 3, 3 limbs (block step + small-ring tail)", lambda k: br_margin(cx, 2048, 1, 7, 3, 3, 3, k, 64), q(k18), rows, log)
        sweep("blind rotation: N=4096, rank 1, block 7, dnum 3, 3 limbs (pipeline block step)", lambda k: br_margin(cx, 4096, 1, 7, 3, 3, 3, k, 32), q(k18), rows, log)
        sweep("blind rotation `big`: N=2^14, rank 1, block 7, dnum 3, 3 limbs (configs[3])", lambda k: br_margin(cx, 16384, 1, 7, 3, 3, 3, k, 16), q(k18), rows, log)
    if "hal" in fam:
        sweep("per-op dft_apply / vmp_apply_dft_to_dft / idft_apply_consume, N=2^16, 8 limbs", lambda k: hal_idft_margin(cx, 65536, 8, k, 4), q(r12_22), rows, log)
        sweep("per-op sequence, N=2^12, 4 limbs (one-pass small-ring transforms)", lambda k: hal_idft_margin(cx, 4096, 4, k, 8), q(list(range(14, 27))), rows, log)
        sweep("per-op sequence, N=64, 2 limbs", lambda k: hal_idft_margin(cx, 64, 2, k, 8), q(list(range(18, 30))), rows, log)
    with open(args.out, "w") as f:
        f.write("| op / shape | rounding kernels | margin by base2k (max abs(x - round(x)); 0.5 = wrong limb) | unsafe from (margin >= 0.25) |\n|---|---|---|---|\n")
        for label, cells, first_unsafe, notes in rows:
            cs = "  ".join(f"{k}: {m:.1e}" if m is not None else f"{k}: refused ({err})" for k, m, err in cells)
            f.write(f"| {label} | {kernels_of(notes)} | {cs} | {first_unsafe if first_unsafe is not None else 'beyond the sweep'} |\n")
    print(open(args.out).read())


if __name__ == "__main__":
    main()
