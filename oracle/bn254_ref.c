/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by or called from the product path
 * (libh2agg.so).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Plain-C CPU restatement of the reference's "pure calculation context" arithmetic:
 *   MockFieldChip  halo2-snark-aggregator-api/src/mock/arith/field.rs:23-146
 *   MockEccChip    halo2-snark-aggregator-api/src/mock/arith/ecc.rs:24-130
 *   ArithEccChip::multi_exp default (same loop)  halo2-snark-aggregator-api/src/arith/ecc.rs:42-60
 *   eval()'s flat tail  halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:189-200
 * The arithmetic those files call lives in halo2curves 0.2.1 (git tag, commit f75ed26c; reference
 * Cargo.lock:1569-1571), which is NOT vendored under /root/reference; its published algorithm is
 * restated here: bn256 Fq/Fr as 4x64-bit Montgomery limbs (R = 2^256), G1 in Jacobian coordinates on
 * y^2 = x^3 + 3, `G1 * Fr` as an MSB-first double-and-add over the 256-bit canonical scalar.
 *
 * PARITY PINNING: "parity unpinned" by reference golden vectors (the reference has none, SURVEY.md
 * §4/§8c, and cannot be built here: Rust + unvendored git deps).  This file is pinned against
 * oracle/bn254.py (independent big-integer affine arithmetic) on the committed fixtures in
 * tests/golden/, the moduli literals in the reference's verifier.sol:40-41,143-144, and the
 * reference's own algebraic identities (five_native_ecc.rs:60-240) — see tests/test_oracle.py.
 *
 * Encodings: Fr/Fq = 32-byte LE canonical; affine = x||y, identity = 64 zero bytes;
 * jacobian = x||y||z, identity z = 0.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;           /* Montgomery form unless stated */
typedef struct { const uint64_t m[4]; uint64_t inv; fe r1; fe r2; } field_t;

static const field_t FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0x87d20782e4866389ULL,
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};
static const field_t FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0xc2e1f593efffffffULL,
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};

static inline int fe_geq(const fe *a, const uint64_t *m) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] > m[i]) return 1;
        if (a->l[i] < m[i]) return 0;
    }
    return 1;
}
static inline void fe_sub_mod_raw(fe *a, const uint64_t *m) {
    u128 b = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - m[i] - (uint64_t)b;
        a->l[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
    return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3];
}
static inline __attribute__((always_inline)) void fe_add(const field_t *F, fe *o, const fe *a, const fe *b) {
    u128 c = 0;
    fe t;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || fe_geq(&t, F->m)) fe_sub_mod_raw(&t, F->m);
    *o = t;
}
static inline __attribute__((always_inline)) void fe_sub(const field_t *F, fe *o, const fe *a, const fe *b) {
    u128 br = 0;
    fe t;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
        t.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t.l[i] + F->m[i];
            t.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    *o = t;
}
static inline void fe_neg(const field_t *F, fe *o, const fe *a) {
    fe z = {{0, 0, 0, 0}};
    fe_sub(F, o, &z, a);
}
/* Montgomery product a*b*R^-1 mod m (CIOS) */
static inline __attribute__((always_inline)) void fe_mul(const field_t *F, fe *o, const fe *a, const fe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t mm = t[0] * F->inv;
        c = (u128)mm * F->m[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)mm * F->m[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&r, F->m)) fe_sub_mod_raw(&r, F->m);
    *o = r;
}
/* One tight, non-inlined copy per field: inlining sixteen CIOS bodies into g1_add made each of them ~2x slower
 * (spills); halo2curves' Fq::mul is a single out-of-line function too. */
static __attribute__((noinline)) void fq_mul(fe *o, const fe *a, const fe *b) { fe_mul(&FQ, o, a, b); }
static __attribute__((noinline)) void fr_mul(fe *o, const fe *a, const fe *b) { fe_mul(&FR, o, a, b); }
static inline void fe_mul_d(const field_t *F, fe *o, const fe *a, const fe *b) {
    if (F == &FQ) fq_mul(o, a, b); else fr_mul(o, a, b);
}
static inline void fe_sqr(const field_t *F, fe *o, const fe *a) { fe_mul_d(F, o, a, a); }
static void fe_pow(const field_t *F, fe *o, const fe *a, const uint64_t e[4]) {
    fe acc = F->r1;
    for (int i = 255; i >= 0; --i) {
        fe_sqr(F, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fe_mul(F, &acc, &acc, a);
    }
    *o = acc;
}
/* Fermat inverse; returns 0 for 0 (callers that restate `invert().unwrap()` check first) */
static void fe_inv(const field_t *F, fe *o, const fe *a) {
    uint64_t e[4] = {F->m[0] - 2, F->m[1], F->m[2], F->m[3]};
    fe_pow(F, o, a, e);
}
static int fe_from_bytes(const field_t *F, fe *o, const uint8_t *b) { /* canonical -> Montgomery */
    fe t;
    memcpy(t.l, b, 32);
    int ok = !fe_geq(&t, F->m);
    if (!ok) { /* reduce leniently so callers can still proceed; status reports non-canonical input */
        while (fe_geq(&t, F->m)) fe_sub_mod_raw(&t, F->m);
    }
    fe_mul(F, o, &t, &F->r2);
    return ok;
}
static void fe_to_bytes(const field_t *F, uint8_t *b, const fe *a) { /* Montgomery -> canonical */
    fe one = {{1, 0, 0, 0}}, t;
    fe_mul(F, &t, a, &one);
    memcpy(b, t.l, 32);
}

/* ------------------------------------------------------------------ G1 (Jacobian) */
typedef struct { fe x, y, z; } g1;
typedef struct { fe x, y; int inf; } g1a;

static void g1_set_inf(g1 *p) {
    memset(p, 0, sizeof *p);
    p->y = FQ.r1;
}
static inline int g1_is_inf(const g1 *p) { return fe_is_zero(&p->z); }
static void g1_from_aff(g1 *o, const g1a *a) {
    if (a->inf) { g1_set_inf(o); return; }
    o->x = a->x; o->y = a->y; o->z = FQ.r1;
}
/* dbl-2009-l (a = 0) */
static void g1_double(g1 *o, const g1 *p) {
    if (g1_is_inf(p)) { g1_set_inf(o); return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_sqr(&FQ, &a, &p->x);
    fe_sqr(&FQ, &b, &p->y);
    fe_sqr(&FQ, &c, &b);
    fe_add(&FQ, &t, &p->x, &b);
    fe_sqr(&FQ, &t, &t);
    fe_sub(&FQ, &t, &t, &a);
    fe_sub(&FQ, &t, &t, &c);
    fe_add(&FQ, &d, &t, &t);
    fe_add(&FQ, &e, &a, &a);
    fe_add(&FQ, &e, &e, &a);
    fe_sqr(&FQ, &f, &e);
    fq_mul(&z3, &p->y, &p->z);
    fe_add(&FQ, &z3, &z3, &z3);
    fe_sub(&FQ, &x3, &f, &d);
    fe_sub(&FQ, &x3, &x3, &d);
    fe_add(&FQ, &c, &c, &c);
    fe_add(&FQ, &c, &c, &c);
    fe_add(&FQ, &c, &c, &c);
    fe_sub(&FQ, &t, &d, &x3);
    fq_mul(&y3, &e, &t);
    fe_sub(&FQ, &y3, &y3, &c);
    o->x = x3; o->y = y3; o->z = z3;
}
/* add-2007-bl with the exceptional cases handled (complete) */
static void g1_add(g1 *o, const g1 *p, const g1 *q) {
    if (g1_is_inf(p)) { *o = *q; return; }
    if (g1_is_inf(q)) { *o = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fe_sqr(&FQ, &z1z1, &p->z);
    fe_sqr(&FQ, &z2z2, &q->z);
    fq_mul(&u1, &p->x, &z2z2);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s1, &p->y, &q->z);
    fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, &q->y, &p->z);
    fq_mul(&s2, &s2, &z1z1);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { g1_double(o, p); return; }
        g1_set_inf(o);
        return;
    }
    fe_sub(&FQ, &h, &u2, &u1);
    fe_add(&FQ, &i, &h, &h);
    fe_sqr(&FQ, &i, &i);
    fq_mul(&j, &h, &i);
    fe_sub(&FQ, &r, &s2, &s1);
    fe_add(&FQ, &r, &r, &r);
    fq_mul(&v, &u1, &i);
    fe_sqr(&FQ, &x3, &r);
    fe_sub(&FQ, &x3, &x3, &j);
    fe_sub(&FQ, &x3, &x3, &v);
    fe_sub(&FQ, &x3, &x3, &v);
    fq_mul(&s1, &s1, &j);
    fe_add(&FQ, &s1, &s1, &s1);
    fe_sub(&FQ, &t, &v, &x3);
    fq_mul(&y3, &r, &t);
    fe_sub(&FQ, &y3, &y3, &s1);
    fe_add(&FQ, &z3, &p->z, &q->z);
    fe_sqr(&FQ, &z3, &z3);
    fe_sub(&FQ, &z3, &z3, &z1z1);
    fe_sub(&FQ, &z3, &z3, &z2z2);
    fq_mul(&z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1_neg(g1 *o, const g1 *p) {
    *o = *p;
    fe_neg(&FQ, &o->y, &p->y);
}
static void g1_add_aff(g1 *o, const g1 *p, const g1a *q) {
    g1 t;
    g1_from_aff(&t, q);
    g1_add(o, p, &t);
}
/* halo2curves `G1 * Fr`: MSB-first double-and-add over the 256-bit canonical repr */
static void g1_scalar_mul(g1 *o, const g1 *p, const uint8_t s[32]) {
    g1 acc;
    g1_set_inf(&acc);
    for (int i = 255; i >= 0; --i) {
        g1_double(&acc, &acc);
        if ((s[i >> 3] >> (i & 7)) & 1) g1_add(&acc, &acc, p);
    }
    *o = acc;
}
static void g1_to_aff(g1a *o, const g1 *p) {
    if (g1_is_inf(p)) { memset(o, 0, sizeof *o); o->inf = 1; return; }
    fe zi, zi2, zi3;
    fe_inv(&FQ, &zi, &p->z);
    fe_sqr(&FQ, &zi2, &zi);
    fq_mul(&zi3, &zi2, &zi);
    fq_mul(&o->x, &p->x, &zi2);
    fq_mul(&o->y, &p->y, &zi3);
    o->inf = 0;
}
static void aff_from_bytes(g1a *o, const uint8_t *b) {
    int z = 1;
    for (int i = 0; i < 64; ++i) z &= (b[i] == 0);
    if (z) { memset(o, 0, sizeof *o); o->inf = 1; return; }
    fe_from_bytes(&FQ, &o->x, b);
    fe_from_bytes(&FQ, &o->y, b + 32);
    o->inf = 0;
}
static void aff_to_bytes(uint8_t *b, const g1a *a) {
    if (a->inf) { memset(b, 0, 64); return; }
    fe_to_bytes(&FQ, b, &a->x);
    fe_to_bytes(&FQ, b + 32, &a->y);
}
static void jac_from_bytes(g1 *o, const uint8_t *b) {
    fe_from_bytes(&FQ, &o->x, b);
    fe_from_bytes(&FQ, &o->y, b + 32);
    fe_from_bytes(&FQ, &o->z, b + 64);
}
static void jac_to_bytes(uint8_t *b, const g1 *p) {
    fe_to_bytes(&FQ, b, &p->x);
    fe_to_bytes(&FQ, b + 32, &p->y);
    fe_to_bytes(&FQ, b + 64, &p->z);
}

/* ================================================================== exported API (C ABI, bytes) */
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_SQR = 3, OP_INV = 4 };

/* which: 0 = Fr, 1 = Fq.  MockFieldChip::{add,sub,mul,square,div} (mock/arith/field.rs:39-55,98-122) */
int oracle_field_batch_op(int which, int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
    const field_t *F = which ? &FQ : &FR;
    for (size_t i = 0; i < n; ++i) {
        fe x, y, z;
        fe_from_bytes(F, &x, a + 32 * i);
        if (b) fe_from_bytes(F, &y, b + 32 * i);
        switch (op) {
        case OP_ADD: fe_add(F, &z, &x, &y); break;
        case OP_SUB: fe_sub(F, &z, &x, &y); break;
        case OP_MUL: fe_mul(F, &z, &x, &y); break;
        case OP_SQR: fe_sqr(F, &z, &x); break;
        case OP_INV:
            if (fe_is_zero(&x)) return 2; /* `invert().unwrap()` panics (mock/arith/field.rs:113) */
            fe_inv(F, &z, &x);
            break;
        default: return 1;
        }
        fe_to_bytes(F, out + 32 * i, &z);
    }
    return 0;
}

/* ArithFieldChip::mul_add_accumulate default (arith/field.rs:68-81): acc = acc*b + v, acc0 = 0 */
int oracle_fr_mul_add_accumulate(const uint8_t *v, size_t n, const uint8_t b[32], uint8_t out[32]) {
    fe acc = {{0, 0, 0, 0}}, bb, x;
    fe_from_bytes(&FR, &bb, b);
    for (size_t i = 0; i < n; ++i) {
        fe_from_bytes(&FR, &x, v + 32 * i);
        fe_mul(&FR, &acc, &acc, &bb);
        fe_add(&FR, &acc, &acc, &x);
    }
    fe_to_bytes(&FR, out, &acc);
    return 0;
}

/* MockEccChip::add / sub (mock/arith/ecc.rs:30-46) on Jacobian operands */
int oracle_g1_batch_add(const uint8_t *a, const uint8_t *b, size_t n, int subtract, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) {
        g1 p, q, r;
        jac_from_bytes(&p, a + 96 * i);
        jac_from_bytes(&q, b + 96 * i);
        if (subtract) g1_neg(&q, &q);
        g1_add(&r, &p, &q);
        jac_to_bytes(out + 96 * i, &r);
    }
    return 0;
}

/* MockEccChip::scalar_mul_constant (mock/arith/ecc.rs:97-104): affine base * scalar */
int oracle_g1_batch_scalar_mul(const uint8_t *bases_aff, const uint8_t *scalars, size_t n, uint8_t *out_jac) {
    for (size_t i = 0; i < n; ++i) {
        g1a a;
        g1 p, r;
        aff_from_bytes(&a, bases_aff + 64 * i);
        g1_from_aff(&p, &a);
        g1_scalar_mul(&r, &p, scalars + 32 * i);
        jac_to_bytes(out_jac + 96 * i, &r);
    }
    return 0;
}

/* MockEccChip::to_value = to_affine (mock/arith/ecc.rs:64-66) */
int oracle_g1_batch_to_affine(const uint8_t *in_jac, size_t n, uint8_t *out_aff) {
    for (size_t i = 0; i < n; ++i) {
        g1 p;
        g1a a;
        jac_from_bytes(&p, in_jac + 96 * i);
        g1_to_aff(&a, &p);
        aff_to_bytes(out_aff + 64 * i, &a);
    }
    return 0;
}

/* THE REFERENCE ALGORITHM (baseline B0): MockEccChip::multi_exp (mock/arith/ecc.rs:106-129) —
 * n independent double-and-add scalar muls, summed left to right, one thread.  n == 0 -> status 3
 * (the reference panics on `acc.unwrap()`). */
int oracle_multi_exp_naive(const uint8_t *bases_aff, const uint8_t *scalars, size_t n, uint8_t out_aff[64]) {
    if (n == 0) return 3;
    g1 acc;
    for (size_t i = 0; i < n; ++i) {
        g1a a;
        g1 p, cur;
        aff_from_bytes(&a, bases_aff + 64 * i);
        g1_from_aff(&p, &a);
        g1_scalar_mul(&cur, &p, scalars + 32 * i);
        if (i == 0) acc = cur; else g1_add(&acc, &acc, &cur);
    }
    g1a r;
    g1_to_aff(&r, &acc);
    aff_to_bytes(out_aff, &r);
    return 0;
}

/* eval()'s flat tail (evaluation.rs:189-200): multi_exp over entries that carry a scalar, then add
 * every scalar-less point. */
int oracle_eval_flat(const uint8_t *pts_aff, const uint8_t *scalars, const uint8_t *has_scalar, size_t n,
                     uint8_t out_aff[64]) {
    g1 acc;
    int have = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!has_scalar[i]) continue;
        g1a a;
        g1 p, cur;
        aff_from_bytes(&a, pts_aff + 64 * i);
        g1_from_aff(&p, &a);
        g1_scalar_mul(&cur, &p, scalars + 32 * i);
        if (!have) { acc = cur; have = 1; } else g1_add(&acc, &acc, &cur);
    }
    if (!have) return 3;
    for (size_t i = 0; i < n; ++i) {
        if (has_scalar[i]) continue;
        g1a a;
        aff_from_bytes(&a, pts_aff + 64 * i);
        g1_add_aff(&acc, &acc, &a);
    }
    g1a r;
    g1_to_aff(&r, &acc);
    aff_to_bytes(out_aff, &r);
    return 0;
}

/* ------------------------------------------------------------------ baseline B1: fair CPU Pippenger
 * (NOT the reference algorithm; BASELINE.md section 3 "B1").  Unsigned c-bit windows, running-sum bucket reduction.
 * Work is cut into (window, point-range) jobs pulled from a shared counter by `nthreads` workers, so that ALL host cores
 * are used whatever the window count (round-1 verdict: one thread per window capped the "fair" baseline at 16 threads on
 * a 256-core host).  Each job owns a bucket set; a window's partial sums over the point ranges are added at the end. */
typedef struct {
    g1a *bases; const uint8_t *scalars; size_t n; int c; int W, S;         /* S point ranges per window */
    g1 *parts;                                                             /* [W][S] partial window sums */
    volatile long *next;                                                   /* shared job counter */
    const uint8_t *bases_aff; volatile long *next_conv;                    /* phase 0: bytes -> Montgomery, in 4096-point blocks */
} pip_shared;

static unsigned get_window(const uint8_t *s, int bit, int c) {
    unsigned v = 0;
    for (int k = 0; k < c; ++k) {
        int b = bit + k;
        if (b < 256) v |= (unsigned)((s[b >> 3] >> (b & 7)) & 1) << k;
    }
    return v;
}
static void *pip_convert(void *arg) {
    pip_shared *J = (pip_shared *)arg;
    const long nblk = (long)((J->n + 4095) / 4096);
    for (;;) {
        const long b = __sync_fetch_and_add(J->next_conv, 1);
        if (b >= nblk) break;
        const size_t lo = (size_t)b * 4096, hi = lo + 4096 < J->n ? lo + 4096 : J->n;
        for (size_t i = lo; i < hi; ++i) aff_from_bytes(&J->bases[i], J->bases_aff + 64 * i);
    }
    return NULL;
}
static void *pip_worker(void *arg) {
    pip_shared *J = (pip_shared *)arg;
    size_t nb = (size_t)1 << J->c;
    g1 *buckets = (g1 *)malloc(nb * sizeof(g1));
    const long njobs = (long)J->W * J->S;
    for (;;) {
        const long job = __sync_fetch_and_add(J->next, 1);
        if (job >= njobs) break;
        const int w = (int)(job / J->S), sidx = (int)(job % J->S);
        const size_t lo = J->n * (size_t)sidx / (size_t)J->S, hi = J->n * (size_t)(sidx + 1) / (size_t)J->S;
        for (size_t b = 0; b < nb; ++b) g1_set_inf(&buckets[b]);
        for (size_t i = lo; i < hi; ++i) {
            unsigned d = get_window(J->scalars + 32 * i, w * J->c, J->c);
            if (d && !J->bases[i].inf) g1_add_aff(&buckets[d], &buckets[d], &J->bases[i]);
        }
        g1 run, sum;
        g1_set_inf(&run);
        g1_set_inf(&sum);
        for (size_t b = nb - 1; b >= 1; --b) {
            g1_add(&run, &run, &buckets[b]);
            g1_add(&sum, &sum, &run);
        }
        J->parts[job] = sum;
    }
    free(buckets);
    return NULL;
}
int oracle_msm_pippenger(const uint8_t *bases_aff, const uint8_t *scalars, size_t n, int c, int nthreads,
                         uint8_t out_aff[64]) {
    if (n == 0) { memset(out_aff, 0, 64); return 0; }
    if (c < 1 || c > 20) return 1;
    if (nthreads < 1) nthreads = 1;
    int W = (254 + c - 1) / c;
    int S = (nthreads + W - 1) / W;                 /* point ranges per window: W * S >= nthreads jobs */
    if ((size_t)S > n) S = (int)n;
    g1a *bases = (g1a *)malloc(n * sizeof(g1a));
    g1 *parts = (g1 *)malloc((size_t)W * S * sizeof(g1));
    volatile long next = 0, next_conv = 0;
    pip_shared sh = {bases, scalars, n, c, W, S, parts, &next, bases_aff, &next_conv};
    if (nthreads > W * S) nthreads = W * S;
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, pip_convert, &sh);   /* every core converts bases */
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, pip_worker, &sh);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    g1 acc;
    g1_set_inf(&acc);
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) g1_double(&acc, &acc);
        for (int sidx = 0; sidx < S; ++sidx) g1_add(&acc, &acc, &parts[(size_t)w * S + sidx]);
    }
    g1a r;
    g1_to_aff(&r, &acc);
    aff_to_bytes(out_aff, &r);
    free(bases); free(parts); free(th);
    return 0;
}

/* ------------------------------------------------------------------ baseline B1': a competent CPU Pippenger
 * (NOT the reference algorithm; the cross-check that keeps the GPU / naive-CPU ratio honest — VERDICT r2 item 7).
 * Signed c-bit digits (half the buckets), buckets in extended-Jacobian XYZZ with MIXED additions of the affine bases
 * (8M + 2S per insertion instead of the Jacobian 7M + 4S + the digit walk), digits recoded once per scalar in a parallel
 * pass, (window, point-range) jobs from a shared counter sized so that every thread gets several, running-sum bucket
 * reduction per job.  Batched-affine bucket additions (the other ~35 %) are not implemented: stated in bench.py's note. */
typedef struct { fe x, y, zz, zzz; } g1z;       /* identity: zz = 0 */
static inline void g1z_set_inf(g1z *p) { memset(p, 0, sizeof *p); }
static inline int g1z_is_inf(const g1z *p) { return fe_is_zero(&p->zz); }
#define QM(o, a, b) fe_mul(&FQ, o, a, b)
#define QA(o, a, b) fe_add(&FQ, o, a, b)
#define QS(o, a, b) fe_sub(&FQ, o, a, b)
static void g1z_double_aff(g1z *o, const fe *x, const fe *y) {      /* mdbl-2008-s-1 */
    fe u, v, w, s, m, t;
    QA(&u, y, y); QM(&v, &u, &u); QM(&w, &u, &v); QM(&s, x, &v);
    QM(&m, x, x); QA(&t, &m, &m); QA(&m, &t, &m);
    QM(&t, &m, &m); QS(&t, &t, &s); QS(&o->x, &t, &s);
    QS(&t, &s, &o->x); QM(&t, &m, &t); QM(&u, &w, y); QS(&o->y, &t, &u);
    o->zz = v; o->zzz = w;
}
static __attribute__((noinline)) void g1z_add_aff(g1z *a, const fe *x2, const fe *y2) {   /* madd-2008-s; (x2, y2) finite; ONE copy: ten inlined products per site would not fit the L1 instruction cache twice */
    if (g1z_is_inf(a)) { a->x = *x2; a->y = *y2; a->zz = FQ.r1; a->zzz = FQ.r1; return; }
    fe u2, s2, p, r, pp, ppp, q, t, v;
    QM(&u2, x2, &a->zz); QM(&s2, y2, &a->zzz);
    QS(&p, &u2, &a->x); QS(&r, &s2, &a->y);
    if (fe_is_zero(&p)) {
        if (fe_is_zero(&r)) g1z_double_aff(a, x2, y2); else g1z_set_inf(a);
        return;
    }
    QM(&pp, &p, &p); QM(&ppp, &p, &pp); QM(&q, &a->x, &pp);
    QM(&t, &r, &r); QS(&t, &t, &ppp); QS(&t, &t, &q); QS(&v, &t, &q);          /* X3 */
    QS(&t, &q, &v); QM(&t, &r, &t); QM(&q, &a->y, &ppp); QS(&a->y, &t, &q);    /* Y3 */
    a->x = v;
    QM(&a->zz, &a->zz, &pp); QM(&a->zzz, &a->zzz, &ppp);
}
static void g1z_add(g1z *a, const g1z *b) {                          /* add-2008-s, complete */
    if (g1z_is_inf(b)) return;
    if (g1z_is_inf(a)) { *a = *b; return; }
    fe u1, u2, s1, s2, p, r, pp, ppp, q, t, v;
    QM(&u1, &a->x, &b->zz); QM(&u2, &b->x, &a->zz); QM(&s1, &a->y, &b->zzz); QM(&s2, &b->y, &a->zzz);
    QS(&p, &u2, &u1); QS(&r, &s2, &s1);
    if (fe_is_zero(&p)) {
        if (!fe_is_zero(&r)) { g1z_set_inf(a); return; }
        fe u, vv, w, s, m;                                           /* dbl-2008-s-1 */
        QA(&u, &a->y, &a->y); QM(&vv, &u, &u); QM(&w, &u, &vv); QM(&s, &a->x, &vv);
        QM(&m, &a->x, &a->x); QA(&t, &m, &m); QA(&m, &t, &m);
        QM(&t, &m, &m); QS(&t, &t, &s); QS(&v, &t, &s);
        QS(&t, &s, &v); QM(&t, &m, &t); QM(&u, &w, &a->y); QS(&a->y, &t, &u);
        a->x = v; QM(&a->zz, &vv, &a->zz); QM(&a->zzz, &w, &a->zzz);
        return;
    }
    QM(&pp, &p, &p); QM(&ppp, &p, &pp); QM(&q, &u1, &pp);
    QM(&t, &r, &r); QS(&t, &t, &ppp); QS(&t, &t, &q); QS(&v, &t, &q);
    QS(&t, &q, &v); QM(&t, &r, &t); QM(&q, &s1, &ppp); QS(&a->y, &t, &q);
    a->x = v;
    QM(&t, &a->zz, &b->zz); QM(&a->zz, &t, &pp);
    QM(&t, &a->zzz, &b->zzz); QM(&a->zzz, &t, &ppp);
}
static void g1z_to_jac(g1 *o, const g1z *p) {   /* (X, Y, ZZ, ZZZ) -> Jacobian with Z = ZZZ / ZZ: X' = X Z^2 / ZZ ... use (X ZZ, Y ZZZ, ZZ): x = X ZZ / ZZ^2, y = Y ZZZ / ZZ^3 */
    if (g1z_is_inf(p)) { g1_set_inf(o); return; }
    QM(&o->x, &p->x, &p->zz);
    QM(&o->y, &p->y, &p->zzz);
    o->z = p->zz;
}
typedef struct {
    const g1a *bases; const int32_t *digits; size_t n; int c, W, S;
    g1 *parts; volatile long *next;
    const uint8_t *scalars; volatile long *next_rec;
    int32_t *digits_out;
} pip2_shared;
/* signed digits of one canonical 256-bit little-endian scalar: d_w in (-2^(c-1), 2^(c-1)], sum d_w 2^(c w) = k */
static void recode_signed(const uint8_t *sc, int c, int W, int32_t *out, size_t stride) {
    uint64_t w[5] = {0, 0, 0, 0, 0};
    memcpy(w, sc, 32);
    int carry = 0;
    const int32_t half = 1 << (c - 1), full = 1 << c;
    for (int i = 0; i < W; ++i) {
        const int bit = i * c;
        int32_t d = 0;
        if (bit < 256) {
            const int q = bit >> 6, r = bit & 63;
            uint64_t v = w[q] >> r;
            if (r + c > 64) v |= w[q + 1] << (64 - r);
            d = (int32_t)(v & (uint64_t)(full - 1));
        }
        d += carry;
        carry = 0;
        if (d > half) { d -= full; carry = 1; }
        out[(size_t)i * stride] = d;
    }
}
static void *pip2_recode(void *arg) {
    pip2_shared *J = (pip2_shared *)arg;
    const long nblk = (long)((J->n + 4095) / 4096);
    for (;;) {
        const long b = __sync_fetch_and_add(J->next_rec, 1);
        if (b >= nblk) break;
        const size_t lo = (size_t)b * 4096, hi = lo + 4096 < J->n ? lo + 4096 : J->n;
        for (size_t i = lo; i < hi; ++i) recode_signed(J->scalars + 32 * i, J->c, J->W, J->digits_out + i, J->n);
    }
    return NULL;
}
static void *pip2_worker(void *arg) {
    pip2_shared *J = (pip2_shared *)arg;
    const size_t nb = ((size_t)1 << (J->c - 1)) + 1;                 /* magnitudes 1 .. 2^(c-1) */
    g1z *buckets = (g1z *)malloc(nb * sizeof(g1z));
    const long njobs = (long)J->W * J->S;
    for (;;) {
        const long job = __sync_fetch_and_add(J->next, 1);
        if (job >= njobs) break;
        const int w = (int)(job / J->S), sidx = (int)(job % J->S);
        const size_t lo = J->n * (size_t)sidx / (size_t)J->S, hi = J->n * (size_t)(sidx + 1) / (size_t)J->S;
        memset(buckets, 0, nb * sizeof(g1z));
        const int32_t *dg = J->digits + (size_t)w * J->n;
        for (size_t i = lo; i < hi; ++i) {
            const int32_t d = dg[i];
            if (!d || J->bases[i].inf) continue;
            if (i + 8 < hi) __builtin_prefetch(&buckets[dg[i + 8] < 0 ? -dg[i + 8] : dg[i + 8]]);
            fe ny;
            const fe *y = &J->bases[i].y;
            if (d < 0) {
                fe_neg(&FQ, &ny, y);
                y = &ny;
            }
            g1z_add_aff(&buckets[d < 0 ? -d : d], &J->bases[i].x, y);
        }
        g1z run, sum;
        g1z_set_inf(&run);
        g1z_set_inf(&sum);
        for (size_t b = nb - 1; b >= 1; --b) {
            g1z_add(&run, &buckets[b]);
            g1z_add(&sum, &run);
        }
        g1z_to_jac(&J->parts[job], &sum);
    }
    free(buckets);
    return NULL;
}
/* jobs_per_thread: (window, range) jobs each thread gets on average (>= 1; 4 balances 16 threads over 17 windows) */
int oracle_msm_pippenger2(const uint8_t *bases_aff, const uint8_t *scalars, size_t n, int c, int nthreads, int jobs_per_thread,
                          uint8_t out_aff[64]) {
    if (n == 0) { memset(out_aff, 0, 64); return 0; }
    if (c < 2 || c > 20) return 1;
    if (nthreads < 1) nthreads = 1;
    if (jobs_per_thread < 1) jobs_per_thread = 1;
    const int W = (255 + c - 1) / c;                                 /* one spare bit for the last carry */
    int S = (nthreads * jobs_per_thread + W - 1) / W;
    if ((size_t)S > n) S = (int)n;
    if (S < 1) S = 1;
    g1a *bases = (g1a *)malloc(n * sizeof(g1a));
    int32_t *digits = (int32_t *)malloc((size_t)W * n * sizeof(int32_t));
    g1 *parts = (g1 *)malloc((size_t)W * S * sizeof(g1));
    volatile long next = 0, next_conv = 0, next_rec = 0;
    pip_shared conv = {bases, scalars, n, c, W, S, NULL, &next, bases_aff, &next_conv};
    pip2_shared sh = {bases, digits, n, c, W, S, parts, &next, scalars, &next_rec, digits};
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, pip_convert, &conv);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, pip2_recode, &sh);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, pip2_worker, &sh);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    g1 acc;
    g1_set_inf(&acc);
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) g1_double(&acc, &acc);
        for (int sidx = 0; sidx < S; ++sidx) g1_add(&acc, &acc, &parts[(size_t)w * S + sidx]);
    }
    g1a r;
    g1_to_aff(&r, &acc);
    aff_to_bytes(out_aff, &r);
    free(bases); free(digits); free(parts); free(th);
    return 0;
}

/* Montgomery constants, exported so tests can pin them against big-integer arithmetic */
void oracle_constants(int which, uint8_t mod[32], uint8_t r1[32], uint8_t r2[32], uint64_t *inv) {
    const field_t *F = which ? &FQ : &FR;
    memcpy(mod, F->m, 32);
    memcpy(r1, F->r1.l, 32);
    memcpy(r2, F->r2.l, 32);
    *inv = F->inv;
}
