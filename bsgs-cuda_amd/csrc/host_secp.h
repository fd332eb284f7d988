// host_secp.h -- host-side secp256k1 arithmetic of the PRODUCT (header-only C++17).
//
// Used by the C-ABI library (giant generator set-up) and by the C++ host (tile dispenser, hit
// resolver).  It plays the role the reference gives to lib/Curve64.pb on the host (GetJob
// 1_9_7File.pb:2077-2092, checkerThread 1_9_7File.pb:3933-4296) but is an independent design:
// Jacobian coordinates, mixed additions and batched normalisation instead of one binary-GCD
// inversion per affine operation (Curve64.pb:2470-2619).  It shares no code with oracle/.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace hs {

typedef unsigned __int128 u128;

struct Fe { uint64_t l[4]; };                     // little-endian limbs, value < 2^256
struct Affine { Fe x, y; bool inf = false; };
struct Jac { Fe x, y, z; bool inf = true; };
typedef Fe Scalar;                                // integers mod n use the same container

static const Fe FE_P = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
static const Fe SC_N = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
static const uint64_t FE_K = 0x1000003D1ULL;

inline Fe fe_from_u64(uint64_t v) { Fe r = {{v, 0, 0, 0}}; return r; }
inline bool fe_is_zero(const Fe &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool fe_equal(const Fe &a, const Fe &b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
inline int fe_cmp(const Fe &a, const Fe &b)
{
    for (int i = 3; i >= 0; i--) { if (a.l[i] < b.l[i]) return -1; if (a.l[i] > b.l[i]) return 1; }
    return 0;
}
inline uint64_t raw_add(Fe &r, const Fe &a, const Fe &b)
{
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
inline uint64_t raw_sub(Fe &r, const Fe &a, const Fe &b)
{
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
    return br;
}
// generic modular add/sub for canonical inputs (used for both p and n)
inline Fe mod_add(const Fe &a, const Fe &b, const Fe &m)
{
    Fe r; uint64_t c = raw_add(r, a, b);
    if (c || fe_cmp(r, m) >= 0) raw_sub(r, r, m);
    return r;
}
inline Fe mod_sub(const Fe &a, const Fe &b, const Fe &m)
{
    Fe r; if (raw_sub(r, a, b)) raw_add(r, r, m);
    return r;
}
inline Fe fe_add(const Fe &a, const Fe &b) { return mod_add(a, b, FE_P); }
inline Fe fe_sub(const Fe &a, const Fe &b) { return mod_sub(a, b, FE_P); }
inline Fe fe_neg(const Fe &a) { return fe_is_zero(a) ? a : mod_sub(FE_P, a, FE_P); }

inline Fe fe_reduce512(const uint64_t w[8])
{
    uint64_t t[5]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)w[4 + i] * FE_K; t[i] = (uint64_t)c; c >>= 64; }
    t[4] = (uint64_t)c;
    Fe r; c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)w[i] + t[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    u128 u = (u128)(t[4] + (uint64_t)c) * FE_K;
    c = (u128)r.l[0] + (uint64_t)u; r.l[0] = (uint64_t)c; c >>= 64;
    c += (u128)r.l[1] + (uint64_t)(u >> 64); r.l[1] = (uint64_t)c; c >>= 64;
    c += r.l[2]; r.l[2] = (uint64_t)c; c >>= 64;
    c += r.l[3]; r.l[3] = (uint64_t)c; c >>= 64;
    if ((uint64_t)c) { Fe k = {{FE_K, 0, 0, 0}}; raw_add(r, r, k); }       // wrapped past 2^256: +K (value is tiny)
    if (fe_cmp(r, FE_P) >= 0) raw_sub(r, r, FE_P);
    return r;
}
inline Fe fe_mul(const Fe &a, const Fe &b)
{
    uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 4; j++) {
        u128 carry = 0;
        for (int i = 0; i < 4; i++) { u128 t = (u128)a.l[i] * b.l[j] + w[i + j] + carry; w[i + j] = (uint64_t)t; carry = t >> 64; }
        w[j + 4] = (uint64_t)carry;
    }
    return fe_reduce512(w);
}
inline Fe fe_sqr(const Fe &a) { return fe_mul(a, a); }
inline Fe fe_sqr_n(Fe a, int n) { for (int i = 0; i < n; i++) a = fe_sqr(a); return a; }
inline Fe fe_inv(const Fe &a)
{   // a^(p-2), run-length addition chain (255 S + 15 M)
    Fe x2 = fe_mul(fe_sqr(a), a), x3 = fe_mul(fe_sqr(x2), a), x6 = fe_mul(fe_sqr_n(x3, 3), x3), x9 = fe_mul(fe_sqr_n(x6, 3), x3);
    Fe x11 = fe_mul(fe_sqr_n(x9, 2), x2), x22 = fe_mul(fe_sqr_n(x11, 11), x11), x44 = fe_mul(fe_sqr_n(x22, 22), x22);
    Fe x88 = fe_mul(fe_sqr_n(x44, 44), x44), x176 = fe_mul(fe_sqr_n(x88, 88), x88), x220 = fe_mul(fe_sqr_n(x176, 44), x44);
    Fe x223 = fe_mul(fe_sqr_n(x220, 3), x3);
    Fe t = fe_mul(fe_sqr_n(x223, 23), x22);
    t = fe_mul(fe_sqr_n(t, 5), a);
    t = fe_mul(fe_sqr_n(t, 3), x2);
    return fe_mul(fe_sqr_n(t, 2), a);
}
inline Fe fe_sqrt(const Fe &a)
{   // a^((p+1)/4): (p+1)/4 = 2^254 - 2^30 - 244 ; plain square-and-multiply (cold path: key parsing)
    Fe e = {{0xFFFFFFFFBFFFFF0CULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x3FFFFFFFFFFFFFFFULL}};
    Fe r = fe_from_u64(1), b = a;
    for (int i = 0; i < 256; i++) { if ((e.l[i / 64] >> (i % 64)) & 1) r = fe_mul(r, b); b = fe_sqr(b); }
    return r;
}

// ---- scalars mod n ---------------------------------------------------------------------------------
inline Scalar sc_add(const Scalar &a, const Scalar &b) { return mod_add(a, b, SC_N); }
inline Scalar sc_sub(const Scalar &a, const Scalar &b) { return mod_sub(a, b, SC_N); }
inline Scalar sc_neg(const Scalar &a) { return fe_is_zero(a) ? a : mod_sub(SC_N, a, SC_N); }
inline Scalar sc_from_u128(u128 v) { Scalar r = {{(uint64_t)v, (uint64_t)(v >> 64), 0, 0}}; return r; }
// a * m for a small enough that the product stays below 2^256 (tile bookkeeping only)
inline Scalar sc_mul_small(const Scalar &a, uint64_t m)
{
    Scalar r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] * m; r.l[i] = (uint64_t)c; c >>= 64; }
    while (fe_cmp(r, SC_N) >= 0) raw_sub(r, r, SC_N);
    return r;
}

// ---- points --------------------------------------------------------------------------------------------
static const Affine G = {{{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}},
                         {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}}, false};

inline Affine affine_neg(const Affine &a) { Affine r = a; r.y = fe_neg(a.y); return r; }
inline Jac to_jac(const Affine &a) { Jac r; r.inf = a.inf; r.x = a.x; r.y = a.y; r.z = fe_from_u64(1); return r; }

inline Jac jac_double(const Jac &p)
{
    if (p.inf || fe_is_zero(p.y)) { Jac r; r.inf = true; return r; }
    Fe a = fe_sqr(p.x), b = fe_sqr(p.y), c = fe_sqr(b);
    Fe d = fe_sub(fe_sqr(fe_add(p.x, b)), fe_add(a, c)); d = fe_add(d, d);
    Fe e = fe_add(fe_add(a, a), a), f = fe_sqr(e);
    Jac r; r.inf = false;
    r.x = fe_sub(f, fe_add(d, d));
    Fe c8 = fe_add(c, c); c8 = fe_add(c8, c8); c8 = fe_add(c8, c8);
    r.y = fe_sub(fe_mul(e, fe_sub(d, r.x)), c8);
    r.z = fe_mul(fe_add(p.y, p.y), p.z);
    return r;
}
inline Jac jac_add_affine(const Jac &p, const Affine &q)
{
    if (q.inf) return p;
    if (p.inf) return to_jac(q);
    Fe z2 = fe_sqr(p.z), u2 = fe_mul(q.x, z2), s2 = fe_mul(q.y, fe_mul(z2, p.z));
    Fe h = fe_sub(u2, p.x), rr = fe_sub(s2, p.y);
    if (fe_is_zero(h)) { if (fe_is_zero(rr)) return jac_double(p); Jac r; r.inf = true; return r; }
    Fe h2 = fe_sqr(h), h3 = fe_mul(h2, h), v = fe_mul(p.x, h2);
    Jac r; r.inf = false;
    r.x = fe_sub(fe_sub(fe_sqr(rr), h3), fe_add(v, v));
    r.y = fe_sub(fe_mul(rr, fe_sub(v, r.x)), fe_mul(p.y, h3));
    r.z = fe_mul(p.z, h);
    return r;
}
inline Affine to_affine(const Jac &p)
{
    Affine r; if (p.inf) { r.inf = true; r.x = r.y = fe_from_u64(0); return r; }
    Fe zi = fe_inv(p.z), zi2 = fe_sqr(zi);
    r.inf = false; r.x = fe_mul(p.x, zi2); r.y = fe_mul(p.y, fe_mul(zi2, zi));
    return r;
}
// batched normalisation: one inversion for the whole vector (Montgomery's trick)
inline std::vector<Affine> batch_to_affine(const std::vector<Jac> &v)
{
    std::vector<Affine> out(v.size());
    std::vector<Fe> pref(v.size());
    Fe acc = fe_from_u64(1);
    for (size_t i = 0; i < v.size(); i++) { pref[i] = acc; if (!v[i].inf) acc = fe_mul(acc, v[i].z); }
    Fe inv = fe_inv(acc);
    for (size_t i = v.size(); i-- > 0;) {
        if (v[i].inf) { out[i].inf = true; out[i].x = out[i].y = fe_from_u64(0); continue; }
        Fe zi = fe_mul(inv, pref[i]);
        inv = fe_mul(inv, v[i].z);
        Fe zi2 = fe_sqr(zi);
        out[i].inf = false; out[i].x = fe_mul(v[i].x, zi2); out[i].y = fe_mul(v[i].y, fe_mul(zi2, zi));
    }
    return out;
}
inline Affine point_add(const Affine &a, const Affine &b) { return to_affine(jac_add_affine(to_jac(a), b)); }
inline Affine point_mul(const Affine &a, const Scalar &k)
{
    Jac r; r.inf = true;
    for (int i = 255; i >= 0; i--) {
        r = jac_double(r);
        if ((k.l[i / 64] >> (i % 64)) & 1) r = jac_add_affine(r, a);
    }
    return to_affine(r);
}
// [1a, 2a, ..., na]
inline std::vector<Affine> multiples(const Affine &a, size_t n)
{
    std::vector<Jac> j(n);
    Jac cur = to_jac(a);
    for (size_t i = 0; i < n; i++) { j[i] = cur; cur = jac_add_affine(cur, a); }
    return batch_to_affine(j);
}
// [(first + i*stride) a], i = 0..count-1
inline std::vector<Affine> strided_multiples(const Affine &a, uint64_t first, uint64_t stride, size_t count)
{
    Affine s = point_mul(a, fe_from_u64(stride));
    std::vector<Jac> j(count);
    Jac cur = to_jac(point_mul(a, fe_from_u64(first)));
    for (size_t i = 0; i < count; i++) { j[i] = cur; cur = jac_add_affine(cur, s); }
    return batch_to_affine(j);
}

// ---- serialisation ---------------------------------------------------------------------------------------
inline Fe fe_from_le(const uint8_t *b) { Fe r; memcpy(r.l, b, 32); return r; }
inline void fe_to_le(const Fe &a, uint8_t *b) { memcpy(b, a.l, 32); }
inline Affine affine_from_le(const uint8_t *x, const uint8_t *y) { Affine r; r.inf = false; r.x = fe_from_le(x); r.y = fe_from_le(y); return r; }
inline void affine_to_le(const Affine &a, uint8_t *x, uint8_t *y) { fe_to_le(a.x, x); fe_to_le(a.y, y); }
inline bool fe_from_hex(Fe &r, const std::string &hex_in)
{
    std::string h = hex_in;
    if (h.size() >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) h = h.substr(2);
    if (h.empty() || h.size() > 64) return false;
    r = fe_from_u64(0);
    for (size_t i = 0; i < h.size(); i++) {
        char c = h[h.size() - 1 - i]; unsigned v;
        if (c >= '0' && c <= '9') v = c - '0'; else if (c >= 'a' && c <= 'f') v = c - 'a' + 10; else if (c >= 'A' && c <= 'F') v = c - 'A' + 10; else return false;
        r.l[i / 16] |= (uint64_t)v << (4 * (i % 16));
    }
    return true;
}
inline std::string fe_to_hex(const Fe &a)
{
    static const char d[] = "0123456789abcdef";
    std::string s(64, '0');
    for (int i = 0; i < 64; i++) s[63 - i] = d[(a.l[i / 16] >> (4 * (i % 16))) & 15];
    return s;
}
inline bool on_curve(const Affine &a) { return fe_equal(fe_sqr(a.y), fe_add(fe_mul(fe_sqr(a.x), a.x), fe_from_u64(7))); }
// 128 hex (x||y), 130 hex (04||x||y) or 66 hex (02/03||x)   (1_9_7File.pb:5006-5018, 274-296)
inline bool parse_pubkey(Affine &out, const std::string &s)
{
    std::string h = s;
    if (h.size() == 130 && h[0] == '0' && h[1] == '4') h = h.substr(2);
    out.inf = false;
    if (h.size() == 128) return fe_from_hex(out.x, h.substr(0, 64)) && fe_from_hex(out.y, h.substr(64));
    if (h.size() == 66 && h[0] == '0' && (h[1] == '2' || h[1] == '3')) {
        if (!fe_from_hex(out.x, h.substr(2))) return false;
        out.y = fe_sqrt(fe_add(fe_mul(fe_sqr(out.x), out.x), fe_from_u64(7)));
        if ((int)(out.y.l[0] & 1) != h[1] - '2') out.y = fe_neg(out.y);
        return true;
    }
    return false;
}
inline std::string compress_pubkey(const Affine &a) { return std::string((a.y.l[0] & 1) ? "03" : "02") + fe_to_hex(a.x); }

}  // namespace hs
