import numpy as np
rs = np.random.RandomState(1)
def f16(x): return np.float16(x)
def fma16(q, a, b):  # single rounding
    return np.float16(np.float64(q) * np.float64(a) + np.float64(b))
N = 20000
fp_exact = fp_f16 = missed = 0
for it in range(N):
    # node: origin p, steps (powers of two), 8 child boxes as bytes
    p = rs.uniform(-1, 1, 3).astype(np.float32)
    step = np.float32(2.0) ** rs.randint(-10, -4, 3)
    lo = rs.randint(0, 200, (8, 3)); hi = lo + rs.randint(1, 56, (8, 3))
    hi = np.minimum(hi, 255)
    # ray: origin near/inside, random direction
    o = (p + step * 128 + rs.uniform(-1, 1, 3) * step * 300).astype(np.float32)
    d = rs.normal(size=3).astype(np.float32); d /= np.linalg.norm(d)
    tmin, tfar = np.float32(1e-4), np.float32(1e5)
    inv = (np.float32(1) / d).astype(np.float32); noi = (-(o * inv)).astype(np.float32)
    a = (step * inv).astype(np.float32); b = (p * inv + noi).astype(np.float32)
    neg = inv < 0
    # exact (float64) test on the quantised boxes
    tl = (lo * a.astype(np.float64) + b); th = (hi * a.astype(np.float64) + b)
    tn = np.where(neg, th, tl).max(1); tf = np.where(neg, tl, th).min(1)
    hit_exact = np.maximum(tn, tmin) <= np.minimum(tf, tfar)
    # f16 path
    nq = np.where(neg, 255.0, 0.0).astype(np.float32)
    cw = np.float32(max((nq * a + b).max(), tmin))
    rinv = np.float32(1) / np.abs(a).max()
    rA, rB = np.float32(rinv * 32768), np.float32(rinv * 0.001953125)
    ncB = np.float32(-cw * rB); pB = np.float32(rB * 0.6)
    B = (b * rB + ncB).astype(np.float32)
    pad = (np.abs(a) * pB + np.float32(2.4e-7)).astype(np.float32)
    A = f16(a * rA); Nn = f16(B - pad); Ff = f16(B + pad)
    hiw = f16(np.float32(tfar * rB + ncB) + np.float32(2.4e-7))
    qn = np.where(neg, hi, lo); qf = np.where(neg, lo, hi)
    qd_n = (qn * 2.0 ** -24); qd_f = (qf * 2.0 ** -24)
    tn16 = np.stack([[fma16(qd_n[s, k], A[k], Nn[k]) for k in range(3)] for s in range(8)]).astype(np.float16)
    tf16 = np.stack([[fma16(qd_f[s, k], A[k], Ff[k]) for k in range(3)] for s in range(8)]).astype(np.float16)
    tnm = np.maximum(tn16.max(1), np.float16(0)); tfm = np.minimum(tf16.min(1), hiw)
    dd = (tfm.astype(np.float16) - tnm.astype(np.float16)).astype(np.float16)
    hit16 = ~np.signbit(dd)
    missed += int((hit_exact & ~hit16).sum())
    fp_exact += int(hit_exact.sum()); fp_f16 += int(hit16.sum())
print("children hit exact %d, f16 %d (x%.3f), missed %d" % (fp_exact, fp_f16, fp_f16 / max(1, fp_exact), missed))
