"""Lane-level numpy model of k_cryst_fused (csrc/ltmi_cryst.hip): the Stockham radix-4 passes with the LDS
swizzle, the staged row pair and the separation of two real rows through lane (-t) mod 64 -- the index
arithmetic of the kernel, checked against np.fft before any HIP was written.  python scripts/cryst_fft_model.py"""
import numpy as np
N=256
def phys(u): return u ^ (5*((u>>4)&3))
t=np.arange(64)
def bfly(u):
    t0=u[0]+u[2]; t1=u[0]-u[2]; t2=u[1]+u[3]; d=u[1]-u[3]; t3=d.imag-1j*d.real
    return [t0+t2,t1+t3,t0-t2,t1-t3]
def fft256(buf, rd0):
    # buf: 256 complex LDS; rd0: read index for pass 1 per lane (then +64r)
    u=[buf[rd0+64*r].copy() for r in range(4)]
    u=bfly(u)
    for r in range(4): buf[phys(4*t+r)]=u[r]
    rd=phys(t)  # phys(t+64r)=phys(t)+64r ?
    for r in range(4): assert (phys(t+64*r)==rd+64*r).all()
    for p in (4,16,64):
        u=[buf[rd+64*r].copy() for r in range(4)]
        k=t&(p-1)
        for r in range(1,4): u[r]=u[r]*np.exp(-2j*np.pi*k*r/(4*p))
        u=bfly(u)
        if p!=64:
            j=((t-k)<<2)+k
            for r in range(4): buf[phys(j+r*p)]=u[r]
    return u  # u[r][t] = Z[t+64r]
rng=np.random.default_rng(0)
z=rng.normal(size=N)+1j*rng.normal(size=N)
buf=z.copy()
u=fft256(buf,t)
Z=np.zeros(N,complex)
for r in range(4): Z[t+64*r]=u[r]
print(abs(Z-np.fft.fft(z)).max())
# staged layout: element idx at idx ^ (2*((idx>>4)&1))
st=np.zeros(N,complex); idx=np.arange(N); st[idx ^ (2*((idx>>4)&1))]=z
u=fft256(st, t ^ (2*((t>>4)&1)))
for r in range(4): Z[t+64*r]=u[r]
print(abs(Z-np.fft.fft(z)).max())
# separation
a=rng.normal(size=N); b=rng.normal(size=N)
Zf=np.fft.fft(a+1j*b)
for K in (10,64,65,71):
  for j in range((K+63)//64):
    k=t+64*j
    # bpermute: dest lane t reads from src lane (-t)&63 ; src supplies reg (t_src==0 ? (4-j)&3 : 3-j)
    src=(-t)&63
    supplied=np.where(t==0, Zf[(t+64*((4-j)&3))], Zf[t+64*(3-j)])  # value each lane supplies
    other=supplied[src]
    assert np.allclose(other, Zf[(N-k)%N])
    zk=Zf[k]
    S=(zk.real+other.real)+1j*(zk.imag-other.imag)
    D=(zk.imag+other.imag)+1j*(other.real-zk.real)
    ok=k<K
    assert np.allclose(S[ok]/2, np.fft.fft(a)[k[ok]]) and np.allclose(D[ok]/2, np.fft.fft(b)[k[ok]])
print('sep ok')
