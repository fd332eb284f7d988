/*
 * oracle/curve64_ref.c -- TEST INFRASTRUCTURE ONLY (the parity oracle).
 * Restates /root/reference/lib/Curve64.pb (cited as C64:line).  See header.
 */
#include "curve64_ref.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;

/* C64:55-59 */
const o_fe O_P  = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
const o_fe O_N  = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
const o_fe O_GX = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
const o_fe O_GY = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};

/* ---- hex: C64:450-473 (m_sethex32 / m_gethex32) -------------------------- */
int o_sethex32(o_fe *r, const char *hex)
{
    size_t n = strlen(hex);
    if (n >= 2 && hex[0] == '0' && (hex[1] == 'x' || hex[1] == 'X')) { hex += 2; n -= 2; }
    if (n > 64) return -1;
    memset(r, 0, sizeof *r);
    for (size_t i = 0; i < n; i++) {
        char c = hex[n - 1 - i];
        unsigned v;
        if (c >= '0' && c <= '9') v = (unsigned)(c - '0');
        else if (c >= 'a' && c <= 'f') v = (unsigned)(c - 'a' + 10);
        else if (c >= 'A' && c <= 'F') v = (unsigned)(c - 'A' + 10);
        else return -1;
        r->l[i / 16] |= (uint64_t)v << (4 * (i % 16));
    }
    return 0;
}

void o_gethex32(char out[65], const o_fe *a)
{
    static const char d[] = "0123456789abcdef";
    for (int i = 0; i < 64; i++)
        out[63 - i] = d[(a->l[i / 16] >> (4 * (i % 16))) & 15];
    out[64] = 0;
}

/* ---- raw ops: C64:700-892 ------------------------------------------------- */
int o_check_nonzero(const o_fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) != 0; }
int o_check_equil(const o_fe *a, const o_fe *b)
{ return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3]; }

int o_check_less_more_equil(const o_fe *a, const o_fe *b)
{
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] < b->l[i]) return 1;
        if (a->l[i] > b->l[i]) return 2;
    }
    return 0;
}

uint64_t o_addX64(o_fe *r, const o_fe *a, const o_fe *b)
{
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}

uint64_t o_subX64(o_fe *r, const o_fe *a, const o_fe *b)
{
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a->l[i] - b->l[i] - br;
        r->l[i] = (uint64_t)t;
        br = (uint64_t)(t >> 64) & 1;
    }
    return br;
}

void o_shrX64(o_fe *a)
{
    a->l[0] = (a->l[0] >> 1) | (a->l[1] << 63);
    a->l[1] = (a->l[1] >> 1) | (a->l[2] << 63);
    a->l[2] = (a->l[2] >> 1) | (a->l[3] << 63);
    a->l[3] >>= 1;
}

void o_andX64(o_fe *r, const o_fe *a, const o_fe *b)
{ for (int i = 0; i < 4; i++) r->l[i] = a->l[i] & b->l[i]; }

/* ---- add/sub mod: C64:893-1036 ------------------------------------------- */
void o_subModX64(o_fe *r, const o_fe *a, const o_fe *b, const o_fe *m)
{   /* borrow -> add modulus once (C64:893-945) */
    o_fe t;
    if (o_subX64(&t, a, b)) o_addX64(&t, &t, m);
    *r = t;
}

void o_addModX64(o_fe *r, const o_fe *a, const o_fe *b, const o_fe *m)
{   /* carry OR (sum strictly greater than m) -> subtract modulus once (C64:947-1036).
       Note the reference's strict '>' : a sum exactly equal to m is left as m. */
    o_fe t;
    uint64_t carry = o_addX64(&t, a, b);
    if (carry || o_check_less_more_equil(&t, m) == 2) o_subX64(&t, &t, m);
    *r = t;
}

/* ---- mul mod p: C64:1038-1437 --------------------------------------------
   16 limb products row by row -> 512 bits; fold high*0x1000003D1 into low
   (512->320->256); final conditional +-p on the (overflow, borrow) pair.      */
void o_mul512(uint64_t r[8], const o_fe *a, const o_fe *b)
{
    memset(r, 0, 8 * sizeof(uint64_t));
    for (int j = 0; j < 4; j++) {
        u128 carry = 0;
        for (int i = 0; i < 4; i++) {
            u128 t = (u128)a->l[i] * b->l[j] + r[i + j] + carry;
            r[i + j] = (uint64_t)t;
            carry = t >> 64;
        }
        r[j + 4] = (uint64_t)carry;
    }
}

static void o_reduce512(o_fe *res, const uint64_t r512[8])
{
    const uint64_t K = 0x1000003D1ULL;
    uint64_t t[5], r[4];
    u128 c = 0;
    /* t[0..4] = high * K  (C64:1330-1380) */
    for (int i = 0; i < 4; i++) { c += (u128)r512[4 + i] * K; t[i] = (uint64_t)c; c >>= 64; }
    t[4] = (uint64_t)c;
    /* low += t[0..3] (C64:1383-1398) */
    c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)r512[i] + t[i]; r[i] = (uint64_t)c; c >>= 64; }
    /* 320 -> 256: (t[4]+carry)*K added at limb 0 (C64:1401-1420) */
    u128 u = (u128)(t[4] + (uint64_t)c) * K;
    c = (u128)r[0] + (uint64_t)u;            r[0] = (uint64_t)c; c >>= 64;
    c += (u128)r[1] + (uint64_t)(u >> 64);   r[1] = (uint64_t)c; c >>= 64;
    c += r[2];                               r[2] = (uint64_t)c; c >>= 64;
    c += r[3];                               r[3] = (uint64_t)c; c >>= 64;
    uint64_t overflow = (uint64_t)c;
    /* final correction (C64:1424-1434) */
    o_fe v = {{r[0], r[1], r[2], r[3]}};
    uint64_t borrow = o_subX64(&v, &v, &O_P);
    if (overflow) { if (!borrow) o_subX64(&v, &v, &O_P); }
    else          { if (borrow)  o_addX64(&v, &v, &O_P); }
    *res = v;
}

void o_mulModX64(o_fe *r, const o_fe *a, const o_fe *b)
{
    uint64_t w[8];
    o_mul512(w, a, b);
    o_reduce512(r, w);
}

/* ---- square mod p: C64:2161-2455 -------------------------------------------
   dedicated squaring: 4 diagonal + 6 cross products (cross terms doubled),
   then the same fold as the multiply.                                          */
void o_squareModX64(o_fe *r, const o_fe *a)
{
    uint64_t w[8] = {0};
    /* cross products a_i*a_j, i<j */
    u128 c;
    u128 t;
    /* row 0 */
    t = (u128)a->l[0] * a->l[1];            w[1] = (uint64_t)t; c = t >> 64;
    t = (u128)a->l[0] * a->l[2] + c;        w[2] = (uint64_t)t; c = t >> 64;
    t = (u128)a->l[0] * a->l[3] + c;        w[3] = (uint64_t)t; w[4] = (uint64_t)(t >> 64);
    /* row 1 */
    t = (u128)a->l[1] * a->l[2] + w[3];     w[3] = (uint64_t)t; c = t >> 64;
    t = (u128)a->l[1] * a->l[3] + w[4] + c; w[4] = (uint64_t)t; w[5] = (uint64_t)(t >> 64);
    /* row 2 */
    t = (u128)a->l[2] * a->l[3] + w[5];     w[5] = (uint64_t)t; w[6] = (uint64_t)(t >> 64);
    /* double */
    w[7] = w[6] >> 63;
    for (int i = 6; i >= 2; i--) w[i] = (w[i] << 1) | (w[i - 1] >> 63);
    w[1] <<= 1;
    /* add diagonals */
    c = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] * a->l[i];
        c += (u128)w[2 * i] + (uint64_t)d;            w[2 * i] = (uint64_t)c;     c >>= 64;
        c += (u128)w[2 * i + 1] + (uint64_t)(d >> 64); w[2 * i + 1] = (uint64_t)c; c >>= 64;
    }
    o_reduce512(r, w);
}

/* ---- modular inverse: C64:2470-2522 (binary extended GCD, "Great Divide") -- */
static void o_modInv_update(o_fe *u, const o_fe *m)
{   /* C64:2457-2468: if odd add modulus (keeping the 257th bit), then >>1 */
    uint64_t carry = 0;
    if (u->l[0] & 1) carry = o_addX64(u, u, m);
    o_shrX64(u);
    if (carry) u->l[3] |= 0x8000000000000000ULL;
}

void o_modInvX64(o_fe *res, const o_fe *inp, const o_fe *m)
{
    if (!o_check_nonzero(inp)) { memset(res, 0, sizeof *res); return; }
    o_fe a = *inp, b = *m, u = {{1, 0, 0, 0}}, v = {{0, 0, 0, 0}};
    int cmp;
    while ((cmp = o_check_less_more_equil(&a, &b)) != 0) {
        if (!(a.l[0] & 1))      { o_shrX64(&a); o_modInv_update(&u, m); }
        else if (!(b.l[0] & 1)) { o_shrX64(&b); o_modInv_update(&v, m); }
        else if (cmp == 2) {
            o_subX64(&a, &a, &b); o_shrX64(&a);
            if (o_check_less_more_equil(&u, &v) == 1) o_addX64(&u, &u, m);
            o_subX64(&u, &u, &v);
            o_modInv_update(&u, m);
        } else {
            o_subX64(&b, &b, &a); o_shrX64(&b);
            if (o_check_less_more_equil(&v, &u) == 1) o_addX64(&v, &v, m);
            o_subX64(&v, &v, &u);
            o_modInv_update(&v, m);
        }
    }
    *res = u;
}

/* ---- EC: C64:2524-2619 ------------------------------------------------------ */
void o_negpt(o_pt *r, const o_pt *a) { r->x = a->x; o_subModX64(&r->y, &O_P, &a->y, &O_P); }

void o_DBLTX64(o_pt *r, const o_pt *a)
{
    o_fe s, dx, tx, ds;
    o_addModX64(&s, &a->y, &a->y, &O_P);
    o_modInvX64(&s, &s, &O_P);
    o_squareModX64(&dx, &a->x);
    o_addModX64(&tx, &dx, &dx, &O_P);
    o_addModX64(&tx, &dx, &tx, &O_P);
    o_mulModX64(&s, &tx, &s);
    o_squareModX64(&ds, &s);
    o_subModX64(&ds, &ds, &a->x, &O_P);
    o_subModX64(&ds, &ds, &a->x, &O_P);
    o_subModX64(&dx, &a->x, &ds, &O_P);
    o_mulModX64(&tx, &s, &dx);
    o_fe ry; o_subModX64(&ry, &tx, &a->y, &O_P);
    r->x = ds; r->y = ry;
}

void o_ADDPTX64(o_pt *r, const o_pt *a, const o_pt *b)
{
    if (o_check_equil(&a->x, &b->x)) { o_DBLTX64(r, a); return; }   /* C64:2561-2562 */
    o_fe s, cx, cy;
    o_subModX64(&s, &a->x, &b->x, &O_P);
    o_modInvX64(&s, &s, &O_P);
    o_subModX64(&cy, &a->y, &b->y, &O_P);
    o_mulModX64(&s, &cy, &s);
    o_squareModX64(&cy, &s);
    o_subModX64(&cy, &cy, &a->x, &O_P);
    o_subModX64(&cx, &cy, &b->x, &O_P);
    o_subModX64(&cy, &a->x, &cx, &O_P);
    o_mulModX64(&cy, &s, &cy);
    o_subModX64(&cy, &cy, &a->y, &O_P);
    r->x = cx; r->y = cy;
}

void o_PTMULX64(o_pt *r, const o_pt *a, const o_fe *k)
{   /* LSB-first double-and-add, "(0,0)" as the empty accumulator (C64:2586-2619) */
    o_fe loc = *k;
    o_pt scale = *a, acc;
    memset(&acc, 0, sizeof acc);
    while (o_check_nonzero(&loc)) {
        if (loc.l[0] & 1) {
            if (!o_check_nonzero(&acc.x) || !o_check_nonzero(&acc.y)) acc = scale;
            else o_ADDPTX64(&acc, &acc, &scale);
        }
        o_DBLTX64(&scale, &scale);
        o_shrX64(&loc);
    }
    *r = acc;
}

void o_YfromX64(o_fe *y, const o_fe *x)
{   /* C64:2656-2682 + DoPowMod C64:2630-2654 */
    o_fe s, seven = {{7, 0, 0, 0}}, one = {{1, 0, 0, 0}}, e, b, acc = {{1, 0, 0, 0}};
    o_mulModX64(&s, x, x);
    o_mulModX64(&s, &s, x);
    o_addModX64(&s, &s, &seven, &O_P);
    o_addX64(&e, &O_P, &one);           /* p+1 does not overflow 256 bits */
    o_shrX64(&e); o_shrX64(&e);
    b = s;
    while (o_check_nonzero(&e)) {
        if (e.l[0] & 1) o_mulModX64(&acc, &acc, &b);
        o_mulModX64(&b, &b, &b);
        o_shrX64(&e);
    }
    *y = acc;
}

/* ---- batched add helpers: C64:2914-3064 -------------------------------------
   arr = n records of 96 bytes {x[32], y[32], diff[32]}.                         */
#define REC_X(arr, i)    ((o_fe *)((arr) + (size_t)(i) * 96))
#define REC_Y(arr, i)    ((o_fe *)((arr) + (size_t)(i) * 96 + 32))
#define REC_D(arr, i)    ((o_fe *)((arr) + (size_t)(i) * 96 + 64))

static void ld(o_fe *d, const void *s) { memcpy(d, s, 32); }
static void st(void *d, const o_fe *s) { memcpy(d, s, 32); }

void o_beginBatchAdd(o_fe *inv_out, size_t n, const o_pt *a, uint8_t *arr)
{
    o_fe s = {{1, 0, 0, 0}}, t, x;
    for (size_t i = 0; i < n; i++) {
        ld(&x, REC_X(arr, i));
        if (o_check_equil(&a->x, &x)) o_addModX64(&t, &a->y, &a->y, &O_P);
        else                          o_subModX64(&t, &a->x, &x, &O_P);
        o_mulModX64(&s, &s, &t);
        st(REC_D(arr, i), &s);
    }
    o_modInvX64(inv_out, &s, &O_P);
}

static void o_batch_one(o_pt *out, const o_pt *a, const o_fe *bx, const o_fe *by, const o_fe *s)
{
    o_fe ny, nx, sl;
    o_subModX64(&ny, &a->y, by, &O_P);
    o_mulModX64(&sl, &ny, s);
    o_squareModX64(&ny, &sl);
    o_subModX64(&ny, &ny, &a->x, &O_P);
    o_subModX64(&nx, &ny, bx, &O_P);
    o_subModX64(&ny, &a->x, &nx, &O_P);
    o_mulModX64(&ny, &ny, &sl);
    o_subModX64(&ny, &ny, &a->y, &O_P);
    out->x = nx; out->y = ny;
}

void o_completeBatchAddWithDouble(uint8_t *newarr, size_t lenline, size_t n,
                                  const o_pt *a, uint8_t *arr, const o_fe *inv_total)
{
    if (n == 0) return;
    o_fe cur = *inv_total, s, t, x, y, d;
    o_pt np;
    for (size_t k = n - 1; ; k--) {
        ld(&x, REC_X(arr, k)); ld(&y, REC_Y(arr, k));
        int eq = o_check_equil(&a->x, &x);
        if (k > 0) {
            ld(&d, REC_D(arr, k - 1));
            o_mulModX64(&s, &cur, &d);
            if (eq) o_addModX64(&t, &a->y, &a->y, &O_P);
            else    o_subModX64(&t, &a->x, &x, &O_P);
            o_mulModX64(&cur, &cur, &t);
        } else s = cur;
        if (eq) o_DBLTX64(&np, a);
        else    o_batch_one(&np, a, &x, &y, &s);
        st(newarr + k * lenline, &np.x);
        st(newarr + k * lenline + 32, &np.y);
        if (k == 0) break;
    }
}

void o_fillarrayN(uint8_t *arr, size_t n, const o_pt *a)
{   /* arr[i] = (i+1)*a, by doubling the filled prefix each round (C64:3033-3064) */
    if (!n) return;
    st(REC_X(arr, 0), &a->x); st(REC_Y(arr, 0), &a->y);
    size_t i = 1;
    while (i < n) {
        size_t k = i;
        if (k + i >= n) k = n - i;
        o_pt last; ld(&last.x, REC_X(arr, i - 1)); ld(&last.y, REC_Y(arr, i - 1));
        o_fe inv;
        o_beginBatchAdd(&inv, k, &last, arr);
        o_completeBatchAddWithDouble(arr + i * 96, 96, k, &last, arr, &inv);
        i += k;
    }
}

void o_mulmod_n_small(o_fe *r, uint64_t a, uint64_t b)
{
    u128 t = (u128)a * b;
    r->l[0] = (uint64_t)t; r->l[1] = (uint64_t)(t >> 64); r->l[2] = r->l[3] = 0;
}
