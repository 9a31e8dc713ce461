"""Accuracy of the inverse Cholesky factor of a 16x16 SPD block: Gaussian elimination on [A | I] in the device order (tile_engine.h:
invCholFactor) against a 4x4-blocked variant (diagonal blocks factored and inverted on their own, panel and trailing update as
matrix products) -- the candidate of DESIGN.md section 9 item 1.  numpy, no GPU."""
import numpy as np
rng=np.random.default_rng(1)
def elim_inv_chol(A):
    # Gaussian elimination on [A | I] in the device's order (rank-1 updates), Li = diag(1/sqrt(pivot)) R
    n=A.shape[0]; A=A.copy(); R=np.eye(n); piv=np.zeros(n)
    for j in range(n):
        d=A[j,j]; piv[j]=d; p=1.0/d
        m=A[:,j]*p; m[:j+1]=0
        aj=A[j,:].copy(); aj[:j+1]=0
        A-=np.outer(m,aj); R-=np.outer(m,R[j,:])
    return R/np.sqrt(piv)[:,None]
def blocked_inv_chol(A,b=4):
    # right-looking blocked Cholesky with explicit inverse of the diagonal blocks; Li by block forward substitution
    n=A.shape[0]; A=A.copy(); L=np.zeros_like(A)
    Linv_diag=[]
    for k in range(0,n,b):
        D=A[k:k+b,k:k+b]
        Ld=np.linalg.cholesky(D)              # 4x4 factor (in registers on the device)
        Ldi=np.linalg.inv(Ld)                 # its inverse (triangular, 4x4)
        L[k:k+b,k:k+b]=Ld; Linv_diag.append(Ldi)
        if k+b<n:
            P=A[k+b:,k:k+b]@Ldi.T             # panel: one MFMA
            L[k+b:,k:k+b]=P
            A[k+b:,k+b:]-=P@P.T               # trailing update: one MFMA
    # inverse factor by blocks
    Li=np.zeros_like(A); nb=n//b
    for i in range(nb):
        Li[i*b:(i+1)*b,i*b:(i+1)*b]=Linv_diag[i]
        for j in range(i):
            acc=np.zeros((b,b))
            for m in range(j,i):
                acc+=L[i*b:(i+1)*b,m*b:(m+1)*b]@Li[m*b:(m+1)*b,j*b:(j+1)*b]
            Li[i*b:(i+1)*b,j*b:(j+1)*b]=-Linv_diag[i]@acc
    return Li
for cond in (1e4,1e8,1e12):
    e1=[];e2=[]
    for t in range(200):
        Q,_=np.linalg.qr(rng.standard_normal((16,16)))
        ev=np.exp(rng.uniform(0,np.log(cond),16)); A=(Q*ev)@Q.T; A=0.5*(A+A.T)
        for f,e in ((elim_inv_chol,e1),(blocked_inv_chol,e2)):
            Li=f(A); e.append(np.abs(Li@A@Li.T-np.eye(16)).max())
    print("cond %.0e  max |Li A Li' - I|: elimination %.2e (median %.2e)   blocked 4x4 %.2e (median %.2e)"%(cond,max(e1),np.median(e1),max(e2),np.median(e2)))
