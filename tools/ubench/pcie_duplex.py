"""PCIe duplex check (tools/ubench): 128 MiB pinned host buffers, H2D alone, D2H alone, both at once on two streams (torch is plumbing)."""
import time
import torch
n = 128 << 20
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def up():
    with torch.cuda.stream(s1): d_a.copy_(h_in, non_blocking=True)
def down():
    with torch.cuda.stream(s2): h_out.copy_(d_b, non_blocking=True)
def both():
    up(); down()
t_up, t_dn, t_both = timed(up), timed(down), timed(both)
gb = n / 1e9
print(f"H2D alone {gb / t_up:.1f} GB/s   D2H alone {gb / t_dn:.1f} GB/s   both at once: {t_both * 1e3:.2f} ms for 2 x 128 MiB = {2 * gb / t_both:.1f} GB/s summed "
      f"(serial would be {(t_up + t_dn) * 1e3:.2f} ms)")
for chunk_mib in (8, 16, 32):
    c = chunk_mib << 20
    def chunks():
        for o in range(0, n, c):
            with torch.cuda.stream(s1): d_a[o:o + c].copy_(h_in[o:o + c], non_blocking=True)
            with torch.cuda.stream(s2): h_out[o:o + c].copy_(d_b[o:o + c], non_blocking=True)
    t = timed(chunks)
    print(f"  in chunks of {chunk_mib} MiB, both directions interleaved: {t * 1e3:.2f} ms = {2 * gb / t:.1f} GB/s summed")
