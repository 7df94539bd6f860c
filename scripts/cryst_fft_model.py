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


# ---- second part: 128 x 128 frames (k_cryst_fused128): two 128-point transforms as one 256-point transform of the
# interleaved sequence; G[kx][y ^ ((kx >> 3) & 1)], column pairs (c, c + 8)
N=128
rng=np.random.default_rng(1)
frame=rng.integers(0,4096,size=(N,N)).astype(float)
mask=(rng.random((N,N))>0.1).astype(float)
x=frame*mask
F=np.fft.rfft2(x)            # (128, 65)
K=65
COL=130
G=np.zeros((K,COL),complex)   # G[kx][y'] , y' = y ^ ((kx>>3)&1)
back=sigma[(-sigma)%64]
tz=[np.exp(2j*np.pi*(sigma+64*k2)/256) for k2 in (0,1)]
for q in range(32):
    rows=[x[4*q+i] for i in range(4)]
    u=[rows[0][2*l]+1j*rows[1][2*l], rows[2][2*l]+1j*rows[3][2*l], rows[0][2*l+1]+1j*rows[1][2*l+1], rows[2][2*l+1]+1j*rows[3][2*l+1]]
    U=fft_core(swapA(u))     # U[k2][l] = W[sigma+64k2]
    Z1=[U[0]+U[2], U[1]+U[3]]                    # 2 Z1[sigma+64 k2]
    Z2=[(U[0]-U[2])*tz[0], (U[1]-U[3])*tz[1]]    # 2 Z2[...]
    for which,Z in enumerate((Z1,Z2)):
        y=4*q+2*which
        give=np.where(l==0, Z[0], Z[1])
        other=give[back]
        zk=Z[0]
        S=(zk.real+other.real)+1j*(zk.imag-other.imag)      # 4 A[kx]
        D=(zk.imag+other.imag)+1j*(other.real-zk.real)      # 4 B[kx]
        for ln in range(64):
            kx=sigma[ln]
            if kx<K:
                sw=(kx>>3)&1
                G[kx][y^sw]=S[ln]; G[kx][(y+1)^sw]=D[ln]
        if K==65:
            ln=0; zk=Z[1][0]        # Z[64], partner itself
            S64=(zk.real+zk.real)+0j*0+1j*(zk.imag-zk.imag); D64=(zk.imag+zk.imag)+1j*(zk.real-zk.real)
            G[64][y^0]=S64; G[64][(y+1)^0]=D64
# check G against row FFTs
R=np.fft.fft(x,axis=1)[:,:K]   # R[y][kx]
for kx in range(K):
    sw=(kx>>3)&1
    col=np.array([G[kx][y^sw] for y in range(N)])
    assert np.allclose(col, 4*R[:,kx]), kx
print('rows ok')
# column stage: pairs (c, c+8)
half=(rng.random((N,K))>0.5).astype(float)
acc=0
pairs=[(c,c+8) for c in range(64) if not (c>>3)&1]+[(64,64)]
for (c1,c2) in pairs:
    mm=l&15; j=l>>4
    u=[]
    for r in range(4):
        y=32*r+2*mm+(j>>1)
        kx=np.where(j&1, c2, c1)
        sw=(kx>>3)&1
        if c1==64: sw=0*sw
        # bank check for 32-lane groups
        units=kx*COL+(y^sw)
        for g in range(2):
            if c1!=c2: assert len(set(units[32*g:32*g+32]%32))==32, (c1,r,g)
        u.append(G[kx, y^sw])
    U=fft_core(u)
    for k2 in (0,1):
        ky=sigma+64*k2
        F1=U[k2]+U[k2+2]; F2=U[k2]-U[k2+2]
        assert np.allclose(F1, 8*F[ky,c1])
        if c2!=c1 or True: assert np.allclose(abs(F2), 8*abs(F[ky,c2]))
        acc+=np.sum(abs(F1)*half[ky,c1])
        if c2!=c1: acc+=np.sum(abs(F2)*half[ky,c2])
ref=np.sum(abs(F)*half)
print('cols ok', acc*0.125, ref)
# bank check of the b64 row-stage stores (16-lane groups, unit mod 16)
for q in (0,5):
  for which in (0,1):
    y=4*q+2*which
    for part in (0,1):
      addr=sigma*COL+((y+part)^((sigma>>3)&1))
      for g in range(4):
        assert len(set(addr[16*g:16*g+16]%16))==16
print('store banks ok')


# ---- third part: 512-point transforms (k_cryst_rows512 / k_cryst_cols512): cf_core on the even and the odd samples +
# one butterfly; separation of two real rows with the partner X[512 - kx] in the lane of 64 - sigma
N=512
rng=np.random.default_rng(2)
def fft512(z):   # z natural (512); returns lo[k2][l] = X[sigma+64k2], hi = X[.. + 256]
    ue=[z[8*l+2*j] for j in range(4)]; uo=[z[8*l+2*j+1] for j in range(4)]
    Ue=fft_core(swapA(ue)); Uo=fft_core(swapA(uo))
    lo=[];hi=[]
    for k2 in range(4):
        w=Uo[k2]*np.exp(-2j*np.pi*(sigma+64*k2)/512)
        lo.append(Ue[k2]+w); hi.append(Ue[k2]-w)
    return lo,hi
z=rng.normal(size=N)+1j*rng.normal(size=N)
lo,hi=fft512(z)
X=np.fft.fft(z)
for k2 in range(4):
    assert np.allclose(lo[k2],X[sigma+64*k2]) and np.allclose(hi[k2],X[sigma+64*k2+256])
print('fft512 ok')
a=rng.normal(size=N); b=rng.normal(size=N)
lo,hi=fft512(a+1j*b)
A=np.fft.fft(a); B=np.fft.fft(b)
back=sigma[(-sigma)%64]
for k2 in range(4):
    give=np.where(l==0, lo[0] if k2==0 else hi[(4-k2)%4], hi[3-k2])
    other=give[back]
    zk=lo[k2]; kx=sigma+64*k2
    S=(zk.real+other.real)+1j*(zk.imag-other.imag); D=(zk.imag+other.imag)+1j*(other.real-zk.real)
    assert np.allclose(S,2*A[kx]) and np.allclose(D,2*B[kx]), k2
# kx = 256
zk=hi[0][0]; S=2*zk.real; D=2*zk.imag
assert np.allclose(S,2*A[256]) and np.allclose(D,2*B[256])
print('sep512 ok')


# ---- fourth part: N = 256 M points (M = 2, 4) as M 256-point transforms + one radix-M butterfly (ch_fft) and the
# separation of two real rows for every block of 64 columns
def bfly_m(us, M):
    if M==2: return [us[0]+us[1], us[0]-us[1]]
    return bfly(us)
def fftN(z, M):
    N=256*M
    sub=[[z[4*M*l+M*j+q] for j in range(4)] for q in range(M)]
    U=[fft_core(swapA(s)) for s in sub]
    X=[[None]*4 for _ in range(M)]
    for k2 in range(4):
        k=sigma+64*k2
        v=[U[q][k2]*np.exp(-2j*np.pi*q*k/N) for q in range(M)]
        o=bfly_m(v,M)
        for m in range(M): X[m][k2]=o[m]
    return X       # X[m][k2][l] = X[sigma+64k2+256m]
rng=np.random.default_rng(3)
back=sigma[(-sigma)%64]
for M in (2,4):
    N=256*M
    z=rng.normal(size=N)+1j*rng.normal(size=N)
    X=fftN(z,M); ref=np.fft.fft(z)
    for m in range(M):
        for k2 in range(4): assert np.allclose(X[m][k2], ref[sigma+64*k2+256*m])
    a=rng.normal(size=N); b=rng.normal(size=N)
    X=fftN(a+1j*b,M); A=np.fft.fft(a); B=np.fft.fft(b)
    for c in range(2*M):               # blocks kx = sigma + 64 c < N/2
        k2,m=c&3,c>>2
        cp=(4*M-c)%(4*M)                # sigma = 0: partner block index
        give=np.where(l==0, X[cp>>2][cp&3], X[M-1-m][3-k2])
        other=give[back]
        zk=X[m][k2]; kx=sigma+64*c
        S=(zk.real+other.real)+1j*(zk.imag-other.imag); D=(zk.imag+other.imag)+1j*(other.real-zk.real)
        assert np.allclose(S,2*A[kx]) and np.allclose(D,2*B[kx]), (M,c)
    zk=X[M//2][0][0]     # kx = N/2: lane 0, block c = 2M
    assert np.allclose(2*zk.real,2*A[N//2]) and np.allclose(2*zk.imag,2*B[N//2])
    print('M',M,'ok')
