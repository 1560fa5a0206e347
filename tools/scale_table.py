"""Table of a 1 -> N GPU run (tools/scale.sh): rate, per-GPU rate and achieved GB/s on algorithmic bytes, efficiency, RCCL ranks."""
import json
import sys


def rows(lines):
    out = []
    for ln in lines:
        ln = ln.strip()
        if not ln:
            continue
        d = json.loads(ln)
        if d.get("skipped") or d.get("failed"):
            out.append((d["n_gpus"], None if d.get("skipped") else {"failed": d.get("rc")}))
            continue
        n = d["n_gpus"]
        per = d.get("per_rank") or []
        gbps = [e.get("pipeline_gbs") for e in per if e.get("pipeline_gbs") is not None]
        cfg = d.get("config") or {}
        par = cfg.get("parallelism", "")
        route = "cabi" if "(cabi)" in par else "torch" if "(torch)" in par else None
        out.append((n, {
            "value": d["value"], "unit": d.get("unit", ""), "per_gpu": d["value"] / n, "ms_per_step": d.get("ms_per_step"),
            "gbps_min": min(gbps) if gbps else None, "gbps_max": max(gbps) if gbps else None,
            "eff": d.get("scaling_efficiency"), "bcast": route, "rccl_ranks": cfg.get("rccl_ranks"),
            "parity": (d.get("parity_sample") or {}).get("ok"),
        }))
    return out


def fmt(table):
    lines = ["| GPUs | whole-job rate | per GPU | ms/step | GB/s per GPU (algorithmic, min - max over ranks) | efficiency | key broadcast | parity |",
             "|---:|---:|---:|---:|---:|---:|---|---|"]
    for n, r in table:
        if r is None:
            lines.append(f"| {n} | skipped (not that many devices on this node) | | | | | | |")
            continue
        if "failed" in r:
            lines.append(f"| {n} | FAILED rc={r['failed']} (see gpurun_out/scale_n{n}.err) | | | | | | |")
            continue
        g = "-" if r["gbps_min"] is None else f"{r['gbps_min']:.0f} - {r['gbps_max']:.0f}"
        e = "-" if r["eff"] is None else f"{r['eff']:.3f}"
        b = "-" if not r["bcast"] else f"{r['bcast']} ({r['rccl_ranks']} ranks)"
        lines.append(f"| {n} | {r['value']:.0f} | {r['per_gpu']:.0f} | {r['ms_per_step']:.3f} | {g} | {e} | {b} | {r['parity']} |")
    return "\n".join(lines)


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        print(fmt(rows(f)))
