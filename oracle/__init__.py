"""ORACLE — TEST INFRASTRUCTURE ONLY.  See oracle/fft64_ref.h.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; poulpy_amd/ never does.
"""
