"""Fit of the bf16-mode GELU (egv_common.h: phi_fast_f): Phi(x) ~ 1 / (1 + exp(-x P(|x|))) with P a polynomial in |x| -- the logit of the
normal distribution function is x times a smooth even function (x^2 / 2 growth in the tails, hence the |x| terms).  Iteratively
re-weighted least squares towards the minimax RELATIVE error of x Phi(x) over |x| <= 6 (floor 2e-3 on the magnitude), degrees 3-6.
The kernel uses degree 5 with -log2(e) folded into the coefficients (max relative error 4.0e-5, absolute 6.6e-6).
usage: python tools/fit_gelu.py"""
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
