"""Algorithmic bytes / flops per unit of the benched operations and the ceilings they are priced against.
Used by bench.py's secondary tools and by tools/roofline_table.py.  Ceilings: HBM 8 000 GB/s (MI355X_MICROARCH.md; 6.3 TB/s
achievable), FP64 68 TFLOP/s (measured VALU FMA peak, profiles/r01_fp64_ubench.txt; MFMA-f64 shares the datapath), L2 -> CU stream
27 000 GB/s for 16-byte coalesced loads of a slice shared by the workgroups of an XCD (profiles/r01_l2_stream.txt)."""
import math
import re

HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 68.0
L2_STREAM_PEAK_GBS = 27000.0


def fft_flops(m: int) -> float:
    """complex FFT of m points (radix-2 count, the usual 5 m log2 m)"""
    return 5.0 * m * math.log2(m)


def glwe_op(n, cols_in, cols_out, a_size, key_size, rows, batch, a_cols=None, extra_in_polys=0):
    """external product / key switch / automorphism: a (a_cols x a_size polys) -> res (cols_out x key_size polys) through a
    rows*cols_in x cols_out*key_size key.  Returns bytes (HBM, algorithmic), flops, key stream bytes (L2 -> CU, per ciphertext
    when every key value is fetched for it alone)."""
    m = n // 2
    a_cols = a_cols if a_cols is not None else cols_in
    npi, npo = cols_in * min(a_size, rows), cols_out * key_size
    key_bytes = rows * cols_in * cols_out * key_size * n * 8
    hbm = (a_cols * a_size + cols_out * key_size + extra_in_polys) * n * 8 + key_bytes / batch
    flops = 8.0 * npi * npo * m + (npi + npo) * fft_flops(m)
    return {"hbm_bytes": hbm, "flops": flops, "key_stream_bytes": key_bytes, "npi": npi, "npo": npo}


def blind_rotation(n, n_lwe, rank, block_size, dnum, brk_size, res_size, batch):
    """CGGI block-binary blind rotation (poulpy-bin-fhe blind_rotation/algorithms/cggi/algorithm.rs:265-368) per rotated ciphertext."""
    m, cols = n // 2, rank + 1
    nb = n_lwe // block_size
    npi, npo = cols * min(dnum, res_size), cols * brk_size
    key_bytes = n_lwe * (dnum * cols) * (cols * brk_size) * n * 8
    flops = nb * (block_size * (8.0 * npi * npo * m + 6.0 * npi * m) + (npi + npo) * fft_flops(m))
    hbm = (n_lwe + 1) * 8 + res_size * cols * n * 8 + (key_bytes + res_size * n * 8) / batch
    return {"hbm_bytes": hbm, "flops": flops, "key_stream_bytes": key_bytes, "npi": npi, "npo": npo, "blocks": nb}


def tensoring(n, rank, size, mode="apply", relin=False, batch=1, one_call=False):
    """GLWE tensoring (poulpy-core operations/glwe.rs:609-913; the convolution half of a CKKS multiplication) per pair, optionally followed
    by glwe_tensor_relinearize (:541-607).  Algorithmic bytes: both operands read (one for `square`), the (rank+1)(rank+2)/2 tensor columns
    written; flops: forward transforms of the operand limbs, the limb convolution of every column pair (Karatsuba cross terms: one product
    per pair, reim4/arithmetic_ref.rs:235-247 upper bound: every limb pair that reaches a result limb), inverse transforms of the tensor."""
    m, cols = n // 2, rank + 1
    tcols = cols * (cols + 1) // 2
    ops = 1 if mode == "square" else 2
    hbm = (ops * cols * size + tcols * size) * n * 8
    nprod = tcols * size * (size + 1) // 2
    flops = (ops * cols * size + tcols * size) * fft_flops(m) + nprod * m * 8.0
    key = 0
    if relin:   # + read of the tensor, write of the GLWE, the key once per call: key switch of rank (rank+1)/2 columns with `size` rows
        pairs = rank * (rank + 1) // 2
        key = size * pairs * cols * size * n * 8
        hbm += (tcols * size + cols * size) * n * 8 + key / batch
        if one_call:   # the tensor is scratch (ckks_mul_into_default): algorithmically neither written nor read - operands in, GLWE out, the key
            hbm = (ops * cols * size + cols * size) * n * 8 + key / batch
        flops += (pairs * size + cols * size) * fft_flops(m) + m * (pairs * size) * (cols * size) * 8.0
    return {"hbm_bytes": hbm, "flops": flops, "key_stream_bytes": key}


def key_share(notes: str) -> int:
    """ciphertexts that share one fetch of a key value, from the library's dispatch notes"""
    best = 1
    for note in notes.split(";"):
        mm = re.search(r"k_br_fused<R0=\d+,CT=(\d+)", note)
        if mm:
            best = max(best, int(mm.group(1)))
        if "k_br_block_lds" in note:
            best = max(best, 8)
        mm = re.search(r"k_mid128r?<CT=(\d+)", note)
        if mm:
            best = max(best, int(mm.group(1)))
    return best


def roofline(value_per_s: float, model: dict, share: int = 1) -> dict:
    """the three ceilings side by side for `value_per_s` units/s"""
    hbm = value_per_s * model["hbm_bytes"] / 1e9
    tf = value_per_s * model["flops"] / 1e12
    l2 = value_per_s * model["key_stream_bytes"] / max(share, 1) / 1e9
    fr = {"hbm": hbm / HBM_PEAK_GBS, "fp64": tf / FP64_PEAK_TFLOPS, "l2_stream": l2 / L2_STREAM_PEAK_GBS}
    # ONE verdict: the largest fraction, and the others that sit within 5 % of it named as a tie (at the metric shape
    # HBM and FP64 are 0.211 vs 0.214 of their peaks - the whole call is equally far from both)
    top = max(fr, key=fr.get)
    ties = sorted(k for k in fr if k != top and fr[top] > 0 and fr[k] >= 0.95 * fr[top])
    return {"nearest_ceiling": top, "tie_with": ties, "frac": fr[top],
            "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr["hbm"], "algorithmic_bytes_per_unit": model["hbm_bytes"]},
            "fp64": {"achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fr["fp64"], "flops_per_unit": model["flops"]},
            "l2_stream": {"achieved": l2, "peak": L2_STREAM_PEAK_GBS, "unit": "GB/s", "frac": fr["l2_stream"],
                          "key_bytes_per_unit": model["key_stream_bytes"], "ciphertexts_per_key_fetch": share}}
