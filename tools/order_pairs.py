#!/usr/bin/env python
"""How many pairs do box-culled sweeps evaluate, per spatial ORDER of the clouds and per block granularity?  (CPU only, numpy.)

C1's clouds at the converged transformation; for sigma2 of EM iterations 12 / 19 / the noise floor the 2^-48 cutoff radius, the pairs
inside the cutoff disc (kd-tree query), and for the Z-curve, the Hilbert curve and the left-aligned kd-tree order of csrc/morton.h the
pairs of all (W owned points) x (G streamed points) blocks whose boxes pass the sweeps' test.  What round 6's change of order rests on:

    sigma2 1.7e-04 (iteration 19): disc 8.4e7 | W=128 G=32: morton 3.67e8 (4.4x)  hilbert 3.10e8 (3.7x)  kd 2.92e8 (3.5x) | kd W=64: 2.23e8 (2.6x)
    sigma2 3.0e-05 (noise floor):  disc 1.5e7 | W=128 G=32: morton 1.93e8 (12.7x) hilbert 1.53e8 (10.1x) kd 1.40e8 (9.2x) | kd W=64: 9.5e7 (6.2x)

    python tools/order_pairs.py [n]
"""
import numpy as np, sys, time
sys.path.insert(0, '/root/repo')
from probreg_amd import synthetic

def morton_keys(p, bits=16):
    lo = p.min(0); ext = (p.max(0)-lo).max()
    q = ((p-lo)/ext*((1<<bits)-1)).astype(np.uint64)
    key = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for k in range(3):
            key |= ((q[:,k]>>np.uint64(b))&np.uint64(1)) << np.uint64(3*b+k)
    return key

def hilbert_keys(p, bits=16):
    # Skilling's transpose algorithm, vectorised
    lo = p.min(0); ext = (p.max(0)-lo).max()
    X = ((p-lo)/ext*((1<<bits)-1)).astype(np.uint64).T.copy()  # [3][n]
    n=3
    M = np.uint64(1)<<np.uint64(bits-1)
    Q = M
    while Q > 1:
        P = Q - np.uint64(1)
        for i in range(n):
            mask = (X[i] & Q) != 0
            # invert
            X[0] = np.where(mask, X[0]^P, X[0])
            # exchange
            t = (X[0]^X[i]) & P
            t = np.where(mask, np.uint64(0), t)
            X[0] ^= t; X[i] ^= t
        Q >>= np.uint64(1)
    for i in range(1,n): X[i] ^= X[i-1]
    t = np.zeros(X.shape[1], np.uint64)
    Q = M
    while Q > 1:
        t = np.where((X[n-1]&Q)!=0, t^(Q-np.uint64(1)), t)
        Q >>= np.uint64(1)
    for i in range(n): X[i] ^= t
    key = np.zeros(X.shape[1], np.uint64)
    for b in range(bits):
        for k in range(3):
            # transpose: X[0] holds most significant of each triple
            key |= ((X[k]>>np.uint64(b))&np.uint64(1)) << np.uint64(3*b+(2-k))
    return key

def boxes(p, w):
    n = len(p)//w*w
    q = p[:n].reshape(-1,w,3)
    return q.min(1), q.max(1)

def count(own, strm, W, G, thr):
    olo, ohi = boxes(own, W); slo, shi = boxes(strm, G)
    tot = 0
    for i in range(0, len(olo), 64):
        gap = np.maximum(np.maximum(olo[i:i+64,None]-shi[None], slo[None]-ohi[i:i+64,None]), 0)
        tot += ((gap**2).sum(-1) <= thr).sum()
    return tot*W*G

def disc_pairs(x, z, thr):
    from scipy.spatial import cKDTree
    t = cKDTree(z)
    return t.query_ball_point(x, np.sqrt(thr), return_length=True).sum()

n = int(sys.argv[1]) if len(sys.argv)>1 else 100000
src, tgt, (r,t,s) = synthetic.rigid_pair(n)
z = src @ r.T + t
for s2 in (2.7e-3, 1.7e-4, 3e-5):
    thr = 48*2*s2*np.log(2)
    dp = disc_pairs(tgt, z, thr)
    print("sigma2 %.1e  radius %.3f disc pairs %.3e" % (s2, np.sqrt(thr), dp))
    for name, kf in (("morton", morton_keys), ("hilbert", hilbert_keys)):
        xo = tgt[np.argsort(kf(tgt), kind='stable')]; zo = z[np.argsort(kf(src), kind='stable')]
        for W,G in ((128,32),(128,16),(64,32),(64,16),(64,8),(32,32)):
            c = count(xo, zo, W, G, thr)
            print("   %-8s W=%3d G=%2d  pairs %.3e  = %.2fx disc" % (name, W, G, c, c/dp))

def kd_order(p, leaf=32):
    idx = np.arange(len(p))
    out = []
    stack = [idx]
    # iterative in-order: use recursion for clarity
    import sys as _s; _s.setrecursionlimit(10000)
    def rec(ix):
        n = len(ix)
        if n <= leaf:
            out.append(ix); return
        q = p[ix]
        ax = np.argmax(q.max(0)-q.min(0))
        nl = ((n+leaf-1)//leaf)      # leaves in this node
        k = (nl//2)*leaf             # left gets a whole number of leaves
        part = np.argpartition(q[:,ax], k-1 if k>0 else 0)
        rec(ix[part[:k]]); rec(ix[part[k:]])
    rec(idx)
    return np.concatenate(out)

print("---- kd order")
for s2 in (2.7e-3, 1.7e-4, 3e-5):
    thr = 48*2*s2*np.log(2)
    dp = disc_pairs(tgt, z, thr)
    xo = tgt[kd_order(tgt)]; zo = z[kd_order(src)]
    for W,G in ((128,32),(128,16),(64,32),(64,16),(64,8),(32,32)):
        c = count(xo, zo, W, G, thr)
        print("   s2 %.1e %-8s W=%3d G=%2d  pairs %.3e  = %.2fx disc" % (s2, "kd", W, G, c, c/dp))
