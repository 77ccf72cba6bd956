#!/usr/bin/env python3
"""Round 6: how many rank lines does the interpolating select (k_select_top / k_select_sdir, bmx_kernels6.h) read per query?
A replay of its search -- directory entry, interpolated guess, header test, up to three secant steps, bisection -- on Bernoulli data
with the directory granularity of configs[3] (CPU, numpy; no GPU).  "current" = the directory as built (an entry names the LINE of its
sampled one); "fb k" = entries that also carry the position inside the line to 2^-k of a line.
Usage: python tools/sim_select_guess.py <density> <lines per directory entry>      e.g.  0.1 85   |   0.01 107"""
import numpy as np, sys
rng = np.random.default_rng(1)
p = float(sys.argv[1]); L = int(sys.argv[2])  # density, target lines per entry -> S
nlines = 400000
ones_per_line = rng.binomial(960, p, nlines).astype(np.int64)
hdr = np.concatenate([[0], np.cumsum(ones_per_line)])  # hdr[j] = ones before line j
total = hdr[-1]
S = 1 << int(np.round(np.log2(L*960*p)))
print("p",p,"S",S,"lines/entry",S/(960*p))
# positions of ones within lines: need bit pos of k-th one in line: approximate as uniform order statistics
nq = 200000
r = rng.integers(1, total+1, nq)  # 1-based
true_line = np.searchsorted(hdr, r, side='left') - 1   # hdr[j] < r <= hdr[j+1]
def frac_pos(k):  # exact fractional line position of 0-based one number k (line + bit/960), bit sampled as order statistic approx
    j = np.searchsorted(hdr, k, side='right') - 1  # hdr[j] <= k < hdr[j+1]
    within = k - hdr[j]; n = ones_per_line[j]
    # position of within-th (0-based) of n uniform points: mean (within+1)/(n+1); sample beta
    b = rng.beta(within+1, np.maximum(n-within,1))
    return j, j + b
nent = (total + S - 1)//S + 1
m_all = np.arange(nent-1)*S
jl, fp = frac_pos(m_all)
jl = np.append(jl, nlines-1); fp = np.append(fp, nlines-1e-9)
def simulate(fb, half):
    idx0 = r-1; m = idx0//S; fr = idx0 % S
    if fb is None:
        lo = jl[m]; hi = jl[m+1]; j = lo + ((hi-lo)*fr)//S
    else:
        U = 1<<fb
        P = np.floor(fp*U).astype(np.int64)
        lo = P[m]>>fb; hi = P[m+1]>>fb
        if half: j = (2*P[m]+1 + (2*(P[m+1]-P[m])*fr)//S) >> (fb+1)
        else: j = (P[m] + ((P[m+1]-P[m])*fr)//S) >> fb
        j = np.clip(j, lo, hi)
    span = hi-lo+1
    reads = np.zeros(nq, dtype=np.int64); open_ = np.ones(nq, bool)
    lo=lo.copy(); hi=hi.copy(); j=j.copy()
    for it in range(40):
        if not open_.any(): break
        reads[open_] += 1
        h = hdr[j]; lt = ones_per_line[j]
        left = open_ & (r <= h); right = open_ & (r > h+lt)
        open_ = left | right
        hi = np.where(left, j-1, hi); lo = np.where(right, j+1, lo)
        away = np.where(left, h-r, r-h-lt-1)
        if it < 3:
            step = 1 + (away*span)//S
            step = np.minimum(step, hi-lo+1)
            nj = np.where(left, j-step, j+step)
            nj = np.clip(nj, lo, hi)
        else: nj = lo + (hi-lo)//2
        j = np.where(open_, nj, j)
    return reads.mean(), (reads==1).mean()
print("current (line-granular):", simulate(None, False))
for fb in (1,2,3,4):
    print("fb",fb, simulate(fb, False), "half:", simulate(fb, True))
