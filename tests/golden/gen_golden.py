#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with PLAIN PYTHON INTEGERS.

This is the independent "second opinion" of SURVEY.md section 8(c): nothing here shares
code with oracle/ (C) or the HIP kernels.  It follows the reference's *formats and
algorithm definition* (reference lines cited there), not its code.

Run from the repo root:  python tests/golden/gen_golden.py          (seconds)
                         python tests/golden/gen_golden.py --cfg1   (adds the -w 20 -htsz 18 digests, ~1 min)
Outputs (committed):
  small_w1024_ht8_t2_b2_p4.json   htGPU/htCPU/G2 images (hex) + tile hit lists + known-key walk
  cfg1_digests.json               sha256 of the -w 2^20 -htsz 18 htCPU/htGPU images and of a G2 image
"""
import hashlib
import json
import os
import struct
import sys

P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
G = (GX, GY)
HERE = os.path.dirname(os.path.abspath(__file__))


def inv(a, m=P):
    return pow(a, -1, m)


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * inv(2 * a[1]) % P
    else:
        lam = (b[1] - a[1]) * inv(b[0] - a[0]) % P
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


def multiples(a, n):
    """[1a, 2a, ..., na] with batched inversion (pure ints)."""
    out = [a]
    while len(out) < n:
        base = out[-1]                       # = len(out) * a
        k = min(len(out) - 1, n - len(out))  # base + out[0..k): never base + base
        if k == 0:
            out.append(add(base, a))
            continue
        ds = [(out[i][0] - base[0]) % P for i in range(k)]
        pref = [1]
        for d in ds:
            pref.append(pref[-1] * d % P)
        iv = inv(pref[-1])
        new = [None] * k
        for i in range(k - 1, -1, -1):
            s = iv * pref[i] % P
            iv = iv * ds[i] % P
            lam = (out[i][1] - base[1]) * s % P
            x = (lam * lam - base[0] - out[i][0]) % P
            new[i] = (x, (lam * (base[0] - x) - base[1]) % P)
        out.extend(new)
    return out[:n]


def key64(pt):
    return pt[0] & 0xFFFFFFFFFFFFFFFF


def pack_tables(keys, htsz):
    """Appendix C: CSR images. keys[i] belongs to position i."""
    items = 1 << htsz
    ents = sorted(((k & (items - 1)) & 0xFFFFFFFF, (k >> 32) & 0xFFFFFFFF, i) for i, k in enumerate(keys))
    offs, kpos = [], 0
    for b in range(items):
        offs.append(kpos)
        while kpos < len(ents) and ents[kpos][0] == b:
            kpos += 1
    hdr = struct.pack("<%dI" % (items + 1), *(offs + [len(keys)]))
    gpu = hdr + struct.pack("<%dI" % len(ents), *(e[1] for e in ents))
    cpu = hdr + b"".join(struct.pack("<II", e[1], e[2]) for e in ents)
    return gpu, cpu


def pack_g2(pts, t, b, p):
    T = t * b
    maxnonce = T * p
    buf = bytearray(64 * maxnonce)
    for i, pt in enumerate(pts):
        tid, j = divmod(i, p)
        for c in range(2):
            v = pt[c]
            for k in range(8):
                word = (v >> (32 * (7 - k))) & 0xFFFFFFFF
                idx = c * 8 * maxnonce + (j * 8 + k) * T + tid
                struct.pack_into("<I", buf, 4 * idx, word)
    return bytes(buf)


def probe(gpu_img, htsz, x):
    items = 1 << htsz
    b = x & (items - 1) & 0xFFFFFFFF
    h = (x >> 32) & 0xFFFFFFFF
    lo, hi = struct.unpack_from("<II", gpu_img, 4 * b)
    base = 4 * (items + 1)
    return any(struct.unpack_from("<I", gpu_img, base + 4 * k)[0] == h for k in range(lo, hi))


def tile_hits(Pt, giants, gpu_img, htsz):
    """Appendix A with v1.9.7 semantics and a CORRECT -Gy (the NEGMODP quirk is not modelled here)."""
    hits = []
    if probe(gpu_img, htsz, Pt[0]):
        hits.append((5, 0xFFFFFFFF))
    for i, g in enumerate(giants):
        if g[0] == Pt[0]:
            s = inv(2 * Pt[1])
            lam = (Pt[1] + g[1]) * s % P
            xm = (lam * lam - Pt[0] - g[0]) % P
            if probe(gpu_img, htsz, xm):
                hits.append((2, i))
            lam = 3 * Pt[0] * Pt[0] * s % P
            xd = (lam * lam - 2 * Pt[0]) % P
            if probe(gpu_img, htsz, xd):
                hits.append((4, i))
            continue
        s = inv(Pt[0] - g[0])
        lam = (Pt[1] + g[1]) * s % P
        xm = (lam * lam - Pt[0] - g[0]) % P
        if probe(gpu_img, htsz, xm):
            hits.append((2, i))
        lam = (Pt[1] - g[1]) * s % P
        xp = (lam * lam - Pt[0] - g[0]) % P
        if probe(gpu_img, htsz, xp):
            hits.append((1, i))
    return sorted(hits, key=lambda h: (h[1], h[0]))


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def small_fixture():
    w, htsz, t, b, p = 1024, 8, 2, 2, 4
    maxnonce = t * b * p
    babies = multiples(G, w)
    assert babies[9] == mul(10) and babies[-1] == mul(w)
    gpu, cpu = pack_tables([key64(q) for q in babies], htsz)
    A = neg(mul(2 * w))
    giants = multiples(A, maxnonce)
    assert giants[-1] == mul(maxnonce, A)
    g2 = pack_g2(giants, t, b, p)
    C = p * w
    gstep = 4 * maxnonce * w
    fx = dict(w=w, htsz=htsz, t=t, b=b, p=p, htgpu=gpu.hex(), htcpu=cpu.hex(), g2=g2.hex(),
              addpubg=["%064x" % A[0], "%064x" % A[1]], center_big="%x" % C, gstep="%x" % gstep, tiles=[])
    # tiles: seeded random P (mostly false positives at this tiny htsz: 32-bit hash vs 1024 babies
    # gives none, so plant true hits by construction) + the known-key walk
    st = 0x5EED
    for n in range(6):
        st, r = splitmix64(st)
        kk = (r % (2**40)) + 5 * w * maxnonce
        Pt = mul(kk)
        fx["tiles"].append(dict(kind="random", k="%x" % kk, px="%064x" % Pt[0], py="%064x" % Pt[1],
                                hits=tile_hits(Pt, giants, gpu, htsz)))
    # planted: P = m*G with m = +-(i+1)*2w +- bb  -> codes 1 / 2 ; m = bb -> code 5 ; m=(i+1)2w -> code 4 path
    for (m, what) in [((3 + 1) * 2 * w + 77, "code1"), (-((5 + 1) * 2 * w) + 300, "code2"),
                      ((7 + 1) * 2 * w - 1024, "code1_edge"), (513, "code5"), (-(2 + 1) * 2 * w - 1, "code2b"),
                      ((6 + 1) * 2 * w, "xequal_plus"), (-(1 + 1) * 2 * w, "xequal_minus")]:
        Pt = mul(m % N)
        fx["tiles"].append(dict(kind=what, k="%x" % (m % N), px="%064x" % Pt[0], py="%064x" % Pt[1],
                                hits=tile_hits(Pt, giants, gpu, htsz)))
    # known-key end-to-end: key 0x1E9AD (reference 1_9_7File.pb:189) searched from start 1
    key, start = 0x1E9AD, 1
    Q = mul(key)
    Qp = add(Q, neg(mul(start)))
    cnt, found = 1, None
    Pt = add(add(Qp, neg(mul(cnt))), neg(mul(C)))
    step_pt = neg(mul(gstep))
    walk = []
    for tile in range(64):
        hs = tile_hits(Pt, giants, gpu, htsz)
        walk.append(dict(cnt="%x" % cnt, px="%064x" % Pt[0], py="%064x" % Pt[1], hits=hs))
        if any(True for _ in hs):
            # resolve: brute-force all sign combos (definition of the contract, Appendix B)
            for code, idx in hs:
                for e1 in (1, -1):
                    for bb in range(1, w + 1):
                        for e2 in (1, -1):
                            g = 0 if code == 5 else (idx + 1) * 2 * w
                            kp = (cnt + C + e1 * g + e2 * bb) % N
                            if kp == (key - start) % N:
                                found = (tile, code, idx)
            if found:
                break
        cnt += gstep
        Pt = add(Pt, step_pt)
    assert found, "known key not reached"
    fx["known_key"] = dict(key="%x" % key, start="%x" % start, qx="%064x" % Q[0], qy="%064x" % Q[1],
                           walk=walk, found_tile=found[0], found_code=found[1], found_idx=found[2])
    with open(os.path.join(HERE, "small_w1024_ht8_t2_b2_p4.json"), "w") as f:
        json.dump(fx, f, indent=0)
    print("small fixture written; found:", found)


def cfg1_digests():
    w, htsz = 1 << 20, 18
    babies = multiples(G, w)
    assert babies[-1] == mul(w)
    gpu, cpu = pack_tables([key64(q) for q in babies], htsz)
    t, b, p = 4, 4, 8
    A = neg(mul(2 * w))
    g2 = pack_g2(multiples(A, t * b * p), t, b, p)
    d = dict(w=w, htsz=htsz,
             htgpu_name="%064x_%d_%d_htGPUv0.BIN" % (GX, w, 1 << htsz), htgpu_size=len(gpu),
             htgpu_sha256=hashlib.sha256(gpu).hexdigest(),
             htcpu_name="%064x_%d_%d_htCPUv0.BIN" % (GX, w, 1 << htsz), htcpu_size=len(cpu),
             htcpu_sha256=hashlib.sha256(cpu).hexdigest(),
             g2_name="%d_%d_%d_%d_g2.BIN" % (t, b, p, w), g2_t=t, g2_b=b, g2_p=p, g2_size=len(g2),
             g2_sha256=hashlib.sha256(g2).hexdigest())
    with open(os.path.join(HERE, "cfg1_digests.json"), "w") as f:
        json.dump(d, f, indent=1)
    print("cfg1 digests written")


if __name__ == "__main__":
    small_fixture()
    if "--cfg1" in sys.argv:
        cfg1_digests()
