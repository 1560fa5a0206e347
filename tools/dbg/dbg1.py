import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from oracle.ref import RefModule
from poulpy_amd.hal import Module
from poulpy_amd.layouts import *
for n in [16384, 32768, 65536]:
    ref, hip = RefModule(n), Module(n)
    rng=np.random.default_rng(n)
    a = VecZnx(n,2,3).fill_uniform(40,rng)
    d = hip.vec_znx_dft_alloc(2,3)
    for c in range(2): hip.vec_znx_dft_apply(1,0,d,c,a,c)
    m=n//2; lg=m.bit_length()-1
    idx=np.arange(m); rev=np.zeros(m,dtype=np.int64)
    for bit in range(lg): rev |= ((idx>>bit)&1)<<(lg-1-bit)
    dr = ref.vec_znx_dft_alloc(2,3)
    for c in range(2): ref.vec_znx_dft_apply(1,0,dr,c,a,c)
    for j in range(3):
        for c in range(2):
            rr=dr.at(c,j); sr=rr[:m]+1j*rr[m:]
            hh=d.at(c,j).view(np.complex128)
            err=np.abs(hh[rev]-sr)
            print(n,'fwd limb',j,'col',c,'maxerr',err.max()/np.abs(sr).max(), 'nz', np.count_nonzero(hh), 'bad', np.count_nonzero(err>1e-6*np.abs(sr).max()))
    big = hip.vec_znx_big_alloc(2,3)
    for c in range(2): hip.vec_znx_idft_apply(big,c,d,c)
    for j in range(3):
        for c in range(2):
            print(n,'inv limb',j,'col',c,'mismatch',np.count_nonzero(big.at(c,j)!=a.at(c,j)))
