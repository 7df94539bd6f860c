import numpy as np
N=256
l=np.arange(64)
def bfly(u):
    t0=u[0]+u[2]; t1=u[0]-u[2]; t2=u[1]+u[3]; d=u[1]-u[3]; t3=d.imag-1j*d.real
    return [t0+t2,t1+t3,t0-t2,t1-t3]
def pl32(A,B):
    A2=A.copy(); B2=B.copy(); A2[32:]=B[:32]; B2[:32]=A[32:]; return A2,B2
def pl16(A,B):
    A2=A.copy(); B2=B.copy()
    for row in (1,3):
        A2[16*row:16*row+16]=B[16*(row-1):16*(row-1)+16]
    for row in (0,2):
        B2[16*row:16*row+16]=A[16*(row+1):16*(row+1)+16]
    return A2,B2
def swapA(R):
    R=[r.copy() for r in R]
    R[0],R[2]=pl32(R[0],R[2]); R[1],R[3]=pl32(R[1],R[3])
    R[0],R[1]=pl16(R[0],R[1]); R[2],R[3]=pl16(R[2],R[3])
    return R
# check swapA == transpose reg <-> lane[5:4]
R=[np.array([100*r+x for x in l],dtype=complex) for r in range(4)]
S=swapA(R)
for r in range(4):
    for x in l:
        lp=(r<<4)|(x&15); rp=x>>4
        assert S[rp][lp]==R[r][x]
print('swapA ok')
def lds_swap(R, sh):   # transpose reg <-> lane bits [sh+1:sh] through a modelled LDS with the xor layout; returns bank conflict info
    lds=np.zeros(256,complex)
    for r in range(4):
        addr=64*r+(l^(r<<sh))
        # write conflict check: 16-lane groups, 16 units(8B) = 32 banks
        for g in range(4):
            a=addr[16*g:16*g+16]%16
            assert len(set(a))==16
        lds[addr]=R[r]
    out=[]
    lb=(l>>sh)&3
    base=64*lb+(l&~(3<<sh))|(lb<<sh)
    for r in range(4):
        addr=base^(r<<sh)
        for g in range(2):
            a=addr[32*g:32*g+32]%32
            assert len(set(a))==32, (sh, r, g)
        out.append(lds[addr])
    return out
S=lds_swap(R,2)
for r in range(4):
    for x in l:
        lp=(x&~12)|(r<<2); rp=(x>>2)&3
        assert S[rp][lp]==R[r][x]
S=lds_swap(R,0)
for r in range(4):
    for x in l:
        lp=(x&~3)|r; rp=x&3
        assert S[rp][lp]==R[r][x]
print('lds swaps ok, conflict-free')
sigma=(l&48)|((l&3)<<2)|((l>>2)&3)
def fft_core(R):   # R in post-swapA layout: lane=16j+m, reg=a2
    m=l&15
    R=bfly(R)
    for c in range(1,4): R[c]=R[c]*np.exp(-2j*np.pi*m*c/64)
    R=lds_swap(R,2)
    R=bfly(R)
    a0=l&3
    for c in range(1,4): R[c]=R[c]*np.exp(-2j*np.pi*a0*c/16)
    R=lds_swap(R,0)
    R=bfly(R)
    R=swapA(R)
    for j in range(1,4): R[j]=R[j]*np.exp(-2j*np.pi*j*sigma/256)
    R=bfly(R)
    return R
rng=np.random.default_rng(0)
z=rng.normal(size=N)+1j*rng.normal(size=N)
R=[z[4*l+j] for j in range(4)]
R=fft_core(swapA(R))
Z=np.zeros(N,complex)
for r in range(4): Z[sigma+64*r]=R[r]
print('row-style fft err', abs(Z-np.fft.fft(z)).max())
# column style: read n = 64 r + 4 m + j with lane = 16 j + m
R=[z[64*r+4*(l&15)+(l>>4)] for r in range(4)]
R=fft_core(R)
for r in range(4): Z[sigma+64*r]=R[r]
print('col-style fft err', abs(Z-np.fft.fft(z)).max())
# separation: partner lane for k = sigma(l)+64j is lane with sigma = (-sigma(l))%64 ; sigma involution
assert (sigma[sigma]==l).all()
back=sigma[(-sigma)%64]   # lane holding k1 = -sigma(l) mod 64
a=rng.normal(size=N); b=rng.normal(size=N); Zf=np.fft.fft(a+1j*b)
Rr=[Zf[sigma+64*r] for r in range(4)]
for K in (10,64,65,71):
  for j in range((K+63)//64):
    k=sigma+64*j
    supplied=np.where(l==0, Rr[(4-j)&3], Rr[3-j])
    other=supplied[back]
    assert np.allclose(other, Zf[(N-k)%N])
print('sep ok')
