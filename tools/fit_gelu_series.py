import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C
def Phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
def phi(x): return np.exp(-0.5*x*x)/np.sqrt(2*np.pi)
def gp(x): return Phi(x)+x*phi(x)
def fit(fun, deg, U, n=3000, iters=30):
    k=np.arange(n); t=np.cos(np.pi*(k+0.5)/n)
    u=np.maximum(np.sqrt((t+1)*U*U/2),1e-7)
    y=(fun(u)-0.5)/u
    w=np.ones_like(t)
    best=None
    for it in range(iters):   # Lawson iteration towards minimax of |u*(r-y)|
        c=C.chebfit(t,y,deg,w=np.sqrt(w)*u)
        e=np.abs(u*(C.chebval(t,c)-y))
        m=e.max()
        if best is None or m<best[0]: best=(m,c)
        w=w*(e/m+1e-3); w/=w.sum()/n
    return C.cheb2poly(best[1])
def evalf32(cp,x,U):
    x=x.astype(np.float32); u=np.clip(x,-U,U).astype(np.float32)
    t=(u*u*np.float32(2/(U*U))-np.float32(1)).astype(np.float32)
    r=np.full_like(t,np.float32(cp[-1]))
    for c in cp[-2::-1]: r=(r*t+np.float32(c)).astype(np.float32)
    return (u*r+np.float32(0.5)).astype(np.float32)
x=np.linspace(-9,9,720001)
for deg in (6,7,8,9):
  for U in (3.8,4.0,4.2,4.4,4.6,4.8,5.0):
    cp=fit(Phi,deg,U); cg=fit(gp,deg,U)
    a=evalf32(cp,x,U).astype(np.float64); g=evalf32(cg,x,U).astype(np.float64)
    print(f"deg {deg} U {U}: Phi err {np.abs(a-Phi(x)).max():.2e}  gelu abs err {np.abs(x*a-x*Phi(x)).max():.2e}  gelu rel err x>0 {np.max(np.abs(a-Phi(x))[x>0]/Phi(x)[x>0]):.2e} | gelu' err {np.abs(g-gp(x)).max():.2e}")
print()
for name,fun,deg,U in (("gelu Phi",Phi,6,3.8),("gelu'",gp,7,4.0)):
    c=fit(fun,deg,U,iters=60)
    a=evalf32(c,x,U).astype(np.float64)
    err=np.abs(a-fun(x)).max()
    print(name,deg,U,"err",err, "scale", 2/(U*U))
    print("{"+", ".join(f"{v:.9e}f" for v in c)+"}")
