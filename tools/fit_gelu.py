import numpy as np
from scipy.special import ndtr, log_ndtr
from scipy.optimize import least_squares
XM=6.0
x = np.linspace(1e-4, XM, 30001)
def gelu_true(x): return x*ndtr(x)
def model(c, x):
    p = np.zeros_like(x)
    for a in c[::-1]:
        p = p*x + a
    return p*x
def errs(c, xs):
    g = model(c, xs)
    with np.errstate(over='ignore'):
        sp = 1/(1+np.exp(-g)); sn = 1/(1+np.exp(g))
    ep = xs*sp - gelu_true(xs)
    en = -xs*sn - gelu_true(-xs)
    return ep, en
def resid(c, xs):
    ep,en = errs(c,xs)
    wp = 1/np.maximum(np.abs(gelu_true(xs)), 2e-3)
    wn = 1/np.maximum(np.abs(gelu_true(-xs)), 2e-3)
    return np.concatenate([ep*wp, en*wn])
g_true = (log_ndtr(x)-log_ndtr(-x))/x
for deg in (3,4,5,6):
    c = np.polyfit(x, g_true, deg)[::-1]
    w = np.ones(2*len(x))
    for it in range(80):
        rr = least_squares(lambda c: resid(c,x)*w, c, method='lm', xtol=1e-15, ftol=1e-15)
        c = rr.x
        e = np.abs(resid(c,x))
        w = w*(1+ 2*e/e.max()); w/=w.mean()
    e = resid(c,x); ep,en = errs(c,x)
    print(deg, 'max rel(floor 2e-3)', np.abs(e).max(), 'abs pos', np.abs(ep).max(), 'neg', np.abs(en).max(), [float(v) for v in c])
