"""Tiny secp256k1 helper on Python integers for Python hosts of the engine (bench.py, examples):
start-up constants only (ADDPUBG, tile stride, tile centres) -- a handful of point operations per run.
Not used on any hot path and not a stand-in for the HIP kernels."""
P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
     0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def mul(k, a=G):
    k %= N
    r = None
    while k:
        if k & 1:
            r = add(r, a)
        a = add(a, a)
        k >>= 1
    return r


def addpubg(w):
    """giant unit ADDPUBG = -(2w)G (1_9_7File.pb:4689-4698)"""
    return neg(mul(2 * w))


def tile_stride(t, b, p, w):
    """(Gstep, PUBADDBIG): Gstep = 4*t*b*p*w keys per tile, PUBADDBIG = -(Gstep)G (1_9_7File.pb:4759-4765)"""
    gstep = 4 * t * b * p * w
    return gstep, neg(mul(gstep))


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)
