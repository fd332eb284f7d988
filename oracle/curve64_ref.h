/*
 * oracle/curve64_ref.h -- TEST INFRASTRUCTURE ONLY (the parity oracle).
 *
 * CPU restatement in plain C of the reference's big-integer / secp256k1 library
 * `lib/Curve64.pb` (PureBasic + inline FASM, module `Curve`).  Nothing under
 * oracle/ is part of the shipped product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it, and only as the checker.
 *
 * Pinned against: the 15 known-answer vectors in the reference's own self-test
 * (Curve64.pb:3067-3397) and plain-Python big-integer
 * fixtures under tests/golden/ (generator committed beside them).
 *
 * Data model = the reference's: every number is a 32-byte little-endian buffer
 * (Curve64.pb:450-461), i.e. 4 x u64 limbs, limb 0 least significant.
 */
#ifndef ORACLE_CURVE64_REF_H
#define ORACLE_CURVE64_REF_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } o_fe;          /* 256-bit LE value            */
typedef struct { o_fe x, y; } o_pt;              /* affine point; (0,0) = none  */

extern const o_fe O_P, O_N, O_GX, O_GY;          /* Curve64.pb:55-59            */

/* hex (Curve64.pb:450-473): 64 hex chars, most significant first, lower case out */
int  o_sethex32(o_fe *r, const char *hex);       /* accepts <=64 digits, returns 0 ok */
void o_gethex32(char out[65], const o_fe *a);

/* raw 256-bit ops (Curve64.pb:700-892) */
int  o_check_nonzero(const o_fe *a);
int  o_check_equil(const o_fe *a, const o_fe *b);
int  o_check_less_more_equil(const o_fe *a, const o_fe *b); /* 0 eq, 1 a<b, 2 a>b */
uint64_t o_addX64(o_fe *r, const o_fe *a, const o_fe *b);   /* returns carry  */
uint64_t o_subX64(o_fe *r, const o_fe *a, const o_fe *b);   /* returns borrow */
void o_shrX64(o_fe *a);
void o_andX64(o_fe *r, const o_fe *a, const o_fe *b);

/* modular (Curve64.pb:893-1036, 1038-1437, 2161-2455, 2470-2522) */
void o_addModX64(o_fe *r, const o_fe *a, const o_fe *b, const o_fe *m);
void o_subModX64(o_fe *r, const o_fe *a, const o_fe *b, const o_fe *m);
void o_mulModX64(o_fe *r, const o_fe *a, const o_fe *b);    /* mod p only */
void o_squareModX64(o_fe *r, const o_fe *a);                /* mod p only */
void o_modInvX64(o_fe *r, const o_fe *a, const o_fe *m);    /* binary GCD */
void o_mul512(uint64_t r[8], const o_fe *a, const o_fe *b); /* for KAT    */

/* EC, affine (Curve64.pb:2524-2682) */
void o_DBLTX64(o_pt *r, const o_pt *a);
void o_ADDPTX64(o_pt *r, const o_pt *a, const o_pt *b);
void o_PTMULX64(o_pt *r, const o_pt *a, const o_fe *k);     /* LSB-first double-and-add */
void o_YfromX64(o_fe *y, const o_fe *x);                    /* (x^3+7)^((p+1)/4) */
void o_negpt(o_pt *r, const o_pt *a);

/* batched add helpers (Curve64.pb:2914-3064). `arr` = 96-byte records {x,y,diff} */
void o_beginBatchAdd(o_fe *inv_out, size_t n, const o_pt *a, uint8_t *arr);
void o_completeBatchAddWithDouble(uint8_t *newarr, size_t lenline, size_t n,
                                  const o_pt *a, uint8_t *arr, const o_fe *inv_total);
void o_fillarrayN(uint8_t *arr, size_t n, const o_pt *a);

/* scalar arithmetic mod n for the host-side model */
void o_mulmod_n_small(o_fe *r, uint64_t a, uint64_t b);      /* a*b (128-bit) as o_fe, no reduction needed */

#ifdef __cplusplus
}
#endif
#endif
