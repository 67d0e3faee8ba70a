// BN254 optimal-ate pairing check, host side of libh2agg.so (one check per aggregation batch: two pairs).
//
// Stands behind the reference's accept / reject signal
//   evaluate_multiopen_proof          halo2-snark-aggregator-api/src/systems/halo2/verify.rs:733-739
//       E::multi_miller_loop(&[(&left_v, &s_g2_prepared), (&right_v, &n_g2_prepared)]).final_exponentiation().is_identity()
//   calc_verify_circuit_final_pair    halo2-snark-aggregator-circuit/src/verify_circuit.rs:175-199 (debug_assert!(success))
//   (on chain: precompile 0x08,       halo2-snark-aggregator-solidity/templates/verifier.sol:5-37)
// whose arithmetic lives in halo2curves 0.2.1 `bn256` (unvendored).  SURVEY.md 8(f) row 4: one pairing per batch, a few
// hundred microseconds of strictly sequential Fq12 arithmetic — there is nothing data-parallel to put on the GPU, so this
// is plain C++ on the host (4 x 64-bit Montgomery limbs, unsigned __int128 products), overlappable with the device work
// of the next batch.
//
// Tower: Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u, Fq12 = Fq6[w]/(w^2 - v).  G2 lives on the D-type
// twist y^2 = x^3 + 3/xi; untwisting (x, y) -> (x w^2, y w^3) makes every line function sparse: l = a*yP + b*xP*w + c*v*w
// (coefficients 0, 3, 4 of the six Fq2 coordinates).  Miller loop over 6x + 2 in homogeneous projective coordinates
// (Costello-Lange-Naehrig 2010 formulas), two Frobenius corrections, final exponentiation = easy part
// (p^6 - 1)(p^2 + 1) then the EXACT hard part (p^4 - p^2 + 1)/r by Scott et al.'s decomposition
// lambda_3 p^3 + lambda_2 p^2 + lambda_1 p + lambda_0 — so the GT element equals the textbook e(P, Q) and can be compared
// coefficient by coefficient with the oracle's flat-basis implementation (tests/test_pairing_capi.py), not only as a boolean.
// This header is included TWICE by h2agg.hip: as namespace `pairing` (portable C++: unsigned __int128 products) and, on x86-64,
// as `pairing_adx` with every function compiled for BMI2 + ADX (PAIRING_NS / PAIRING_ADX set by the includer): there fq_mul is
// the "no-carry" CIOS over mulx with two interleaved carry chains (the top limb of p is below 2^62, so the running sum never
// needs a fifth word): 16.9 instead of 23.5 ns per product in a throughput loop, 26.6 instead of 37.5 in a dependent chain
// (Xeon 2.1 GHz), and every Fq12 operation above it inlines against that.  h2agg.hip picks the namespace once per process from
// the CPU's feature bits; the results are the same integers.
#include <stdint.h>
#include <string.h>
#if defined(PAIRING_ADX)
#include <immintrin.h>
#endif

#include <memory>
#include <vector>

#ifndef PAIRING_NS
#define PAIRING_NS pairing
#endif

namespace h2agg {
namespace PAIRING_NS {

typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------------------------- Fq
struct Fq {
    uint64_t l[4];
};
static const uint64_t FQ_MOD[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FQ_INV = 0x87d20782e4866389ull;   // -p^-1 mod 2^64

static inline bool fq_geq_mod(const uint64_t* a) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > FQ_MOD[i]) return true;
        if (a[i] < FQ_MOD[i]) return false;
    }
    return true;
}
static inline void fq_sub_mod(uint64_t* a) {
    u128 b = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a[i] - FQ_MOD[i] - (uint64_t)b;
        a[i] = (uint64_t)t;
        b = (t >> 64) & 1;
    }
}
// a + b mod p and a - b mod p for values < p, branch-free (the tower spends as many additions as multiplications: a
// mispredicted compare-and-subtract per addition was ~40 % of the Miller loop's time)
static inline Fq fq_add(const Fq& a, const Fq& b) {
    uint64_t s[4], d[4];
    unsigned char cy = 0, bw = 0;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const u128 v = (u128)a.l[i] + b.l[i] + cy;
        s[i] = (uint64_t)v;
        cy = (unsigned char)(v >> 64);
    }
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const u128 v = (u128)s[i] - FQ_MOD[i] - bw;
        d[i] = (uint64_t)v;
        bw = (unsigned char)((v >> 64) & 1);
    }
    const uint64_t m = (uint64_t)0 - (uint64_t)(cy | !bw);   // sum >= p: take the difference
    Fq r;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) r.l[i] = (d[i] & m) | (s[i] & ~m);
    return r;
}
static inline Fq fq_sub(const Fq& a, const Fq& b) {
    uint64_t d[4];
    unsigned char bw = 0, cy = 0;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const u128 v = (u128)a.l[i] - b.l[i] - bw;
        d[i] = (uint64_t)v;
        bw = (unsigned char)((v >> 64) & 1);
    }
    const uint64_t m = (uint64_t)0 - (uint64_t)bw;   // negative: add p back
    Fq r;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const u128 v = (u128)d[i] + (FQ_MOD[i] & m) + cy;
        r.l[i] = (uint64_t)v;
        cy = (unsigned char)(v >> 64);
    }
    return r;
}
static inline Fq fq_zero() { return Fq{{0, 0, 0, 0}}; }
static inline bool fq_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline bool fq_eq(const Fq& a, const Fq& b) {
    return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3];
}
static inline Fq fq_neg(const Fq& a) { return fq_is_zero(a) ? a : fq_sub(fq_zero(), a); }
static inline Fq fq_dbl(const Fq& a) { return fq_add(a, a); }
// Montgomery product a*b/2^256 mod p (CIOS)
#if defined(PAIRING_ADX)
static inline Fq fq_mul(const Fq& a, const Fq& b) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, A, C, hi, lo, m;
    unsigned char c1, c2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // t += a[i] * b   (chain c2 carries the products' high halves, chain c1 the sum into t)
        lo = _mulx_u64(a.l[i], b.l[0], &hi); c1 = _addcarry_u64(0, lo, t0, &t0); A = hi;
        lo = _mulx_u64(a.l[i], b.l[1], &hi); c2 = _addcarry_u64(0, lo, A, &lo); A = hi; c1 = _addcarry_u64(c1, lo, t1, &t1);
        lo = _mulx_u64(a.l[i], b.l[2], &hi); c2 = _addcarry_u64(c2, lo, A, &lo); A = hi; c1 = _addcarry_u64(c1, lo, t2, &t2);
        lo = _mulx_u64(a.l[i], b.l[3], &hi); c2 = _addcarry_u64(c2, lo, A, &lo); A = hi; c1 = _addcarry_u64(c1, lo, t3, &t3);
        _addcarry_u64(c2, A, 0, &A);
        _addcarry_u64(c1, A, 0, &A);                     // A: the word above t3 (no fifth word: p < 2^254)
        // t = (t + m p) / 2^64
        m = t0 * FQ_INV;
        lo = _mulx_u64(m, FQ_MOD[0], &hi); c2 = _addcarry_u64(0, lo, t0, &lo); C = hi;
        lo = _mulx_u64(m, FQ_MOD[1], &hi); c2 = _addcarry_u64(c2, C, lo, &lo); C = hi; c1 = _addcarry_u64(0, lo, t1, &t0);
        lo = _mulx_u64(m, FQ_MOD[2], &hi); c2 = _addcarry_u64(c2, C, lo, &lo); C = hi; c1 = _addcarry_u64(c1, lo, t2, &t1);
        lo = _mulx_u64(m, FQ_MOD[3], &hi); c2 = _addcarry_u64(c2, C, lo, &lo); C = hi; c1 = _addcarry_u64(c1, lo, t3, &t2);
        _addcarry_u64(c2, C, 0, &C);
        _addcarry_u64(c1, C, A, &t3);
    }
    Fq r = {{t0, t1, t2, t3}};
    if (fq_geq_mod(r.l)) fq_sub_mod(r.l);
    return r;
}
#else
static inline Fq fq_mul(const Fq& a, const Fq& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a.l[i] * b.l[j] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * FQ_INV;
        c = (u128)m * FQ_MOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * FQ_MOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fq r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fq_geq_mod(r.l)) fq_sub_mod(r.l);
    return r;
}
#endif
static inline Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }

struct FqConsts {
    Fq one, r2;
    FqConsts() {
        // 2^256 mod p and 2^512 mod p by repeated modular doubling (no transcribed constants to get wrong)
        Fq x = {{1, 0, 0, 0}};
        for (int i = 0; i < 256; ++i) x = fq_add(x, x);
        one = x;
        for (int i = 0; i < 256; ++i) x = fq_add(x, x);
        r2 = x;
    }
};
static inline const FqConsts& fqc() {
    static const FqConsts c;
    return c;
}
static inline Fq fq_one() { return fqc().one; }
// canonical little-endian 32 bytes <-> Montgomery; returns false if the integer is >= p
static inline bool fq_from_bytes(const uint8_t* b, Fq& out) {
    Fq t;
    memcpy(t.l, b, 32);
    if (fq_geq_mod(t.l)) return false;
    out = fq_mul(t, fqc().r2);
    return true;
}
static inline void fq_to_bytes(const Fq& a, uint8_t* b) {
    const Fq one_int = {{1, 0, 0, 0}};
    const Fq t = fq_mul(a, one_int);
    memcpy(b, t.l, 32);
}
static inline Fq fq_from_u64(uint64_t v) {
    const Fq t = {{v, 0, 0, 0}};
    return fq_mul(t, fqc().r2);
}
// a^e for a 256-bit little-endian exponent
static inline Fq fq_pow(const Fq& a, const uint64_t e[4]) {
    Fq acc = fq_one();
    for (int i = 255; i >= 0; --i) {
        acc = fq_sqr(acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = fq_mul(acc, a);
    }
    return acc;
}
static inline Fq fq_inv(const Fq& a) {   // Fermat; a != 0
    uint64_t e[4] = {FQ_MOD[0] - 2, FQ_MOD[1], FQ_MOD[2], FQ_MOD[3]};
    return fq_pow(a, e);
}

// ---------------------------------------------------------------------------------------------- Fq2
struct Fq2 {
    Fq c0, c1;
};
static inline Fq2 f2_zero() { return Fq2{fq_zero(), fq_zero()}; }
static inline Fq2 f2_one() { return Fq2{fq_one(), fq_zero()}; }
static inline bool f2_is_zero(const Fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
static inline bool f2_eq(const Fq2& a, const Fq2& b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
static inline Fq2 f2_add(const Fq2& a, const Fq2& b) { return Fq2{fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
static inline Fq2 f2_sub(const Fq2& a, const Fq2& b) { return Fq2{fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
static inline Fq2 f2_neg(const Fq2& a) { return Fq2{fq_neg(a.c0), fq_neg(a.c1)}; }
static inline Fq2 f2_dbl(const Fq2& a) { return f2_add(a, a); }
static inline Fq2 f2_conj(const Fq2& a) { return Fq2{a.c0, fq_neg(a.c1)}; }
static inline Fq2 f2_mul(const Fq2& a, const Fq2& b) {
    const Fq t0 = fq_mul(a.c0, b.c0), t1 = fq_mul(a.c1, b.c1);
    const Fq t2 = fq_mul(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1));
    return Fq2{fq_sub(t0, t1), fq_sub(fq_sub(t2, t0), t1)};
}
static inline Fq2 f2_sqr(const Fq2& a) {
    const Fq t = fq_mul(a.c0, a.c1);
    return Fq2{fq_mul(fq_add(a.c0, a.c1), fq_sub(a.c0, a.c1)), fq_dbl(t)};
}
static inline Fq2 f2_mul_fq(const Fq2& a, const Fq& k) { return Fq2{fq_mul(a.c0, k), fq_mul(a.c1, k)}; }
static inline Fq2 f2_inv(const Fq2& a) {
    const Fq d = fq_inv(fq_add(fq_sqr(a.c0), fq_sqr(a.c1)));
    return Fq2{fq_mul(a.c0, d), fq_neg(fq_mul(a.c1, d))};
}
// * xi = 9 + u:  (9 a0 - a1) + (9 a1 + a0) u
static inline Fq2 f2_mul_xi(const Fq2& a) {
    const Fq a8_0 = fq_dbl(fq_dbl(fq_dbl(a.c0))), a8_1 = fq_dbl(fq_dbl(fq_dbl(a.c1)));
    return Fq2{fq_sub(fq_add(a8_0, a.c0), a.c1), fq_add(fq_add(a8_1, a.c1), a.c0)};
}
static inline Fq2 f2_halve(const Fq2& a) {
    static const Fq half = fq_inv(fq_from_u64(2));
    return f2_mul_fq(a, half);
}
static inline Fq2 f2_pow(const Fq2& a, const std::vector<uint64_t>& e) {
    Fq2 acc = f2_one();
    for (int i = (int)e.size() * 64 - 1; i >= 0; --i) {
        acc = f2_sqr(acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = f2_mul(acc, a);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------- Fq6
struct Fq6 {
    Fq2 c0, c1, c2;
};
static inline Fq6 f6_zero() { return Fq6{f2_zero(), f2_zero(), f2_zero()}; }
static inline Fq6 f6_one() { return Fq6{f2_one(), f2_zero(), f2_zero()}; }
static inline Fq6 f6_add(const Fq6& a, const Fq6& b) { return Fq6{f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
static inline Fq6 f6_sub(const Fq6& a, const Fq6& b) { return Fq6{f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; }
static inline Fq6 f6_neg(const Fq6& a) { return Fq6{f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; }
static inline Fq6 f6_mul_v(const Fq6& a) { return Fq6{f2_mul_xi(a.c2), a.c0, a.c1}; }   // * v
static inline Fq6 f6_mul(const Fq6& a, const Fq6& b) {
    const Fq2 t0 = f2_mul(a.c0, b.c0), t1 = f2_mul(a.c1, b.c1), t2 = f2_mul(a.c2, b.c2);
    const Fq2 c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a.c1, a.c2), f2_add(b.c1, b.c2)), t1), t2)));
    const Fq2 c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b.c0, b.c1)), t0), t1), f2_mul_xi(t2));
    const Fq2 c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c2), f2_add(b.c0, b.c2)), t0), t2), t1);
    return Fq6{c0, c1, c2};
}
static inline Fq6 f6_sqr(const Fq6& a) { return f6_mul(a, a); }
static inline Fq6 f6_inv(const Fq6& a) {
    const Fq2 c0 = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    const Fq2 c1 = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    const Fq2 c2 = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    const Fq2 t = f2_inv(f2_add(f2_mul(a.c0, c0), f2_mul_xi(f2_add(f2_mul(a.c2, c1), f2_mul(a.c1, c2)))));
    return Fq6{f2_mul(c0, t), f2_mul(c1, t), f2_mul(c2, t)};
}

// ---------------------------------------------------------------------------------------------- Fq12
struct Fq12 {
    Fq6 c0, c1;
};
static inline Fq12 f12_one() { return Fq12{f6_one(), f6_zero()}; }
static inline Fq12 f12_mul(const Fq12& a, const Fq12& b) {
    const Fq6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
    const Fq6 c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), t0), t1);
    return Fq12{f6_add(t0, f6_mul_v(t1)), c1};
}
static inline Fq12 f12_sqr(const Fq12& a) {
    // (a0 + a1 w)^2 = (a0^2 + v a1^2) + 2 a0 a1 w, with a0^2 + v a1^2 = (a0 + a1)(a0 + v a1) - (1 + v) a0 a1
    const Fq6 ab = f6_mul(a.c0, a.c1);
    const Fq6 c0 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_v(a.c1))), ab), f6_mul_v(ab));
    return Fq12{c0, f6_add(ab, ab)};
}
static inline Fq12 f12_conj(const Fq12& a) { return Fq12{a.c0, f6_neg(a.c1)}; }
static inline Fq12 f12_inv(const Fq12& a) {
    const Fq6 t = f6_inv(f6_sub(f6_sqr(a.c0), f6_mul_v(f6_sqr(a.c1))));
    return Fq12{f6_mul(a.c0, t), f6_neg(f6_mul(a.c1, t))};
}
static inline bool f12_is_one(const Fq12& a) {
    return f2_eq(a.c0.c0, f2_one()) && f2_is_zero(a.c0.c1) && f2_is_zero(a.c0.c2) && f2_is_zero(a.c1.c0) &&
           f2_is_zero(a.c1.c1) && f2_is_zero(a.c1.c2);
}
// a * (b0 + b1 v)
static inline Fq6 f6_mul_by_01(const Fq6& a, const Fq2& b0, const Fq2& b1) {
    const Fq2 t0 = f2_mul(a.c0, b0), t1 = f2_mul(a.c1, b1);
    const Fq2 c1 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b0, b1)), t0), t1);
    return Fq6{f2_add(t0, f2_mul_xi(f2_mul(a.c2, b1))), c1, f2_add(t1, f2_mul(a.c2, b0))};
}
// f * (a + b w + c v w): the sparse line element (coordinates 0, 3, 4) — 13 Fq2 products instead of 18
static inline Fq12 f12_mul_by_034(const Fq12& f, const Fq2& a, const Fq2& b, const Fq2& c) {
    const Fq6 t0 = Fq6{f2_mul(f.c0.c0, a), f2_mul(f.c0.c1, a), f2_mul(f.c0.c2, a)};   // f0 * l0
    const Fq6 t1 = f6_mul_by_01(f.c1, b, c);                                           // f1 * l1
    const Fq6 c1 = f6_sub(f6_sub(f6_mul_by_01(f6_add(f.c0, f.c1), f2_add(a, b), c), t0), t1);
    return Fq12{f6_add(t0, f6_mul_v(t1)), c1};
}

// Frobenius coefficients gamma_i = xi^(i (p - 1) / 6), i = 1..5, computed once (no transcribed constants)
struct FrobConsts {
    Fq2 g[6];
    FrobConsts() {
        // (p - 1) / 6 as a 256-bit integer: long division of the limbs by 6
        uint64_t pm1[4] = {FQ_MOD[0] - 1, FQ_MOD[1], FQ_MOD[2], FQ_MOD[3]};
        std::vector<uint64_t> e(4);
        u128 rem = 0;
        for (int i = 3; i >= 0; --i) {
            const u128 cur = (rem << 64) | pm1[i];
            e[i] = (uint64_t)(cur / 6);
            rem = cur % 6;
        }
        const Fq2 xi = {fq_from_u64(9), fq_one()};
        g[0] = f2_one();
        g[1] = f2_pow(xi, e);
        for (int i = 2; i < 6; ++i) g[i] = f2_mul(g[i - 1], g[1]);
    }
};
static inline const FrobConsts& frob() {
    static const FrobConsts c;
    return c;
}
// f^p: conjugate every Fq2 coordinate, coordinate of v^j w^k picks up gamma_(2j + k)
static inline Fq12 f12_frobenius(const Fq12& a) {
    const FrobConsts& F = frob();
    Fq12 r;
    r.c0.c0 = f2_conj(a.c0.c0);
    r.c0.c1 = f2_mul(f2_conj(a.c0.c1), F.g[2]);
    r.c0.c2 = f2_mul(f2_conj(a.c0.c2), F.g[4]);
    r.c1.c0 = f2_mul(f2_conj(a.c1.c0), F.g[1]);
    r.c1.c1 = f2_mul(f2_conj(a.c1.c1), F.g[3]);
    r.c1.c2 = f2_mul(f2_conj(a.c1.c2), F.g[5]);
    return r;
}
// a^2 for a in the cyclotomic subgroup G_{Phi_6(p^2)} (everything after the easy part of the final exponentiation): Granger and
// Scott's squaring — Fq12 seen as three Fq4 = Fq2[y]/(y^2 - xi) pairs (c0.c0, c1.c1), (c1.c0, c0.c2), (c0.c1, c1.c2); 6 Fq2
// products instead of 12.  Checked against f12_sqr on cyclotomic inputs by tests/test_pairing_capi.py (the GT element of every
// pairing goes through it).
static inline Fq12 f12_cyclotomic_sqr(const Fq12& a) {
    auto fp4_sqr = [](const Fq2& x, const Fq2& y, Fq2& t0, Fq2& t1) {   // (x + y Y)^2 = t0 + t1 Y,  Y^2 = xi
        const Fq2 xy = f2_mul(x, y);
        t0 = f2_sub(f2_sub(f2_mul(f2_add(x, y), f2_add(x, f2_mul_xi(y))), xy), f2_mul_xi(xy));
        t1 = f2_dbl(xy);
    };
    Fq2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(a.c0.c0, a.c1.c1, t0, t1);
    fp4_sqr(a.c1.c0, a.c0.c2, t2, t3);
    fp4_sqr(a.c0.c1, a.c1.c2, t4, t5);
    auto three_minus_two = [](const Fq2& t, const Fq2& z) {   // 3 t - 2 z
        const Fq2 d = f2_sub(t, z);
        return f2_add(f2_dbl(d), t);
    };
    auto three_plus_two = [](const Fq2& t, const Fq2& z) {    // 3 t + 2 z
        const Fq2 d = f2_add(t, z);
        return f2_add(f2_dbl(d), t);
    };
    Fq12 r;
    r.c0.c0 = three_minus_two(t0, a.c0.c0);
    r.c1.c1 = three_plus_two(t1, a.c1.c1);
    r.c1.c0 = three_plus_two(f2_mul_xi(t5), a.c1.c0);
    r.c0.c2 = three_minus_two(t4, a.c0.c2);
    r.c0.c1 = three_minus_two(t2, a.c0.c1);
    r.c1.c2 = three_plus_two(t3, a.c1.c2);
    return r;
}
static inline Fq12 f12_pow_x(const Fq12& a) {   // a^x, x = 0x44e992b44a6909f1; `a` in the cyclotomic subgroup
    const uint64_t x = 0x44e992b44a6909f1ull;
    Fq12 acc = a;
    for (int i = 61; i >= 0; --i) {   // bit 62 is the top bit
        acc = f12_cyclotomic_sqr(acc);
        if ((x >> i) & 1) acc = f12_mul(acc, a);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------- G2 and lines
struct G2Affine {
    Fq2 x, y;
    bool inf;
};
struct G2Proj {
    Fq2 x, y, z;
};
static inline Fq2 twist_b() {   // 3 / xi
    static const Fq2 b = f2_mul_fq(f2_inv(Fq2{fq_from_u64(9), fq_one()}), fq_from_u64(3));
    return b;
}
static inline bool g2_on_curve(const G2Affine& q) {
    if (q.inf) return true;
    return f2_eq(f2_sqr(q.y), f2_add(f2_mul(f2_sqr(q.x), q.x), twist_b()));
}
struct Line {
    Fq2 a, b, c;   // l = a*yP + b*xP*w + c*v*w
};
static inline Line g2_double_step(G2Proj& r) {
    const Fq2 a = f2_halve(f2_mul(r.x, r.y));
    const Fq2 b = f2_sqr(r.y), c = f2_sqr(r.z);
    const Fq2 e = f2_mul(twist_b(), f2_add(f2_dbl(c), c));
    const Fq2 f = f2_add(f2_dbl(e), e);
    const Fq2 g = f2_halve(f2_add(b, f));
    const Fq2 h = f2_sub(f2_sqr(f2_add(r.y, r.z)), f2_add(b, c));
    const Fq2 i = f2_sub(e, b);
    const Fq2 j = f2_sqr(r.x);
    const Fq2 e2 = f2_sqr(e);
    r.x = f2_mul(a, f2_sub(b, f));
    r.y = f2_sub(f2_sqr(g), f2_add(f2_dbl(e2), e2));
    r.z = f2_mul(b, h);
    return Line{f2_neg(h), f2_add(f2_dbl(j), j), i};
}
static inline Line g2_add_step(G2Proj& r, const G2Affine& q) {
    const Fq2 theta = f2_sub(r.y, f2_mul(q.y, r.z));
    const Fq2 lambda = f2_sub(r.x, f2_mul(q.x, r.z));
    const Fq2 c = f2_sqr(theta), d = f2_sqr(lambda);
    const Fq2 e = f2_mul(lambda, d), f = f2_mul(r.z, c), g = f2_mul(r.x, d);
    const Fq2 h = f2_sub(f2_add(e, f), f2_dbl(g));
    r.x = f2_mul(lambda, h);
    r.y = f2_sub(f2_mul(theta, f2_sub(g, h)), f2_mul(e, r.y));
    r.z = f2_mul(r.z, e);
    const Fq2 j = f2_sub(f2_mul(theta, q.x), f2_mul(lambda, q.y));
    return Line{lambda, f2_neg(theta), j};
}
static inline void ell(Fq12& f, const Line& l, const Fq& px, const Fq& py) {
    f = f12_mul_by_034(f, f2_mul_fq(l.a, py), f2_mul_fq(l.b, px), l.c);
}
// r * Q == identity ?  (subgroup membership of a twist point; double-and-add in projective coordinates through the same
// step functions, the line values are discarded)
static inline bool g2_in_subgroup(const G2Affine& q) {
    if (q.inf) return true;
    static const uint64_t RMOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    // (r - 1) * Q must equal -Q
    uint64_t e[4] = {RMOD[0] - 1, RMOD[1], RMOD[2], RMOD[3]};
    G2Proj r = {q.x, q.y, f2_one()};
    int top = 255;
    while (!((e[top / 64] >> (top % 64)) & 1)) --top;
    for (int i = top - 1; i >= 0; --i) {
        g2_double_step(r);
        if ((e[i / 64] >> (i % 64)) & 1) g2_add_step(r, q);   // never hits r == +-q before the end for a point of order r
    }
    if (f2_is_zero(r.z)) return false;
    // affine x = X/Z, y = Y/Z must be (q.x, -q.y)
    return f2_eq(r.x, f2_mul(q.x, r.z)) && f2_eq(r.y, f2_neg(f2_mul(q.y, r.z)));
}

struct G1Affine {
    Fq x, y;
    bool inf;
};
static inline bool g1_on_curve(const G1Affine& p) {
    if (p.inf) return true;
    return fq_eq(fq_sqr(p.y), fq_add(fq_mul(fq_sqr(p.x), p.x), fq_from_u64(3)));
}

// prod_i f_{6x+2, Q_i}(P_i) * Frobenius lines — `multi_miller_loop`
static inline Fq12 multi_miller_loop(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs) {
    const uint64_t lo = 0x9d797039be763ba8ull;   // 6x + 2 = 0x1_9d797039be763ba8 (65 bits)
    std::vector<size_t> live;
    std::vector<G2Proj> rs;
    for (size_t i = 0; i < ps.size(); ++i)
        if (!ps[i].inf && !qs[i].inf) {
            live.push_back(i);
            rs.push_back(G2Proj{qs[i].x, qs[i].y, f2_one()});
        }
    Fq12 f = f12_one();
    for (int bit = 63; bit >= 0; --bit) {   // the top bit (bit 64) is the initial R = Q
        f = f12_sqr(f);
        for (size_t k = 0; k < live.size(); ++k) ell(f, g2_double_step(rs[k]), ps[live[k]].x, ps[live[k]].y);
        if ((lo >> bit) & 1)
            for (size_t k = 0; k < live.size(); ++k) ell(f, g2_add_step(rs[k], qs[live[k]]), ps[live[k]].x, ps[live[k]].y);
    }
    const FrobConsts& F = frob();
    for (size_t k = 0; k < live.size(); ++k) {
        const G2Affine& q = qs[live[k]];
        // pi(Q) = (conj(x) gamma_2, conj(y) gamma_3); pi^2(Q) by applying it twice; the second correction uses -pi^2(Q)
        G2Affine q1 = {f2_mul(f2_conj(q.x), F.g[2]), f2_mul(f2_conj(q.y), F.g[3]), false};
        G2Affine q2 = {f2_mul(f2_conj(q1.x), F.g[2]), f2_neg(f2_mul(f2_conj(q1.y), F.g[3])), false};
        ell(f, g2_add_step(rs[k], q1), ps[live[k]].x, ps[live[k]].y);
        ell(f, g2_add_step(rs[k], q2), ps[live[k]].x, ps[live[k]].y);
    }
    return f;
}

// The line coefficients of a FIXED G2 point, in the order the Miller loop consumes them (the reference's `G2Prepared`,
// verify.rs:733-737: s_g2_prepared / n_g2_prepared).  [s]_2 and -[1]_2 of one ParamsKZG come back with every batch: with the
// lines cached, a pair costs its sparse multiplications only.
struct G2Prepared {
    std::vector<Line> lines;
    bool inf = true;
};
static inline G2Prepared g2_prepare(const G2Affine& q) {
    G2Prepared out;
    out.inf = q.inf;
    if (q.inf) return out;
    const uint64_t lo = 0x9d797039be763ba8ull;
    G2Proj r = {q.x, q.y, f2_one()};
    for (int bit = 63; bit >= 0; --bit) {
        out.lines.push_back(g2_double_step(r));
        if ((lo >> bit) & 1) out.lines.push_back(g2_add_step(r, q));
    }
    const FrobConsts& F = frob();
    G2Affine q1 = {f2_mul(f2_conj(q.x), F.g[2]), f2_mul(f2_conj(q.y), F.g[3]), false};
    G2Affine q2 = {f2_mul(f2_conj(q1.x), F.g[2]), f2_neg(f2_mul(f2_conj(q1.y), F.g[3])), false};
    out.lines.push_back(g2_add_step(r, q1));
    out.lines.push_back(g2_add_step(r, q2));
    return out;
}
// the same product of line evaluations as multi_miller_loop, from prepared lines
static inline Fq12 multi_miller_loop_prepared(const std::vector<G1Affine>& ps, const std::vector<const G2Prepared*>& qs) {
    const uint64_t lo = 0x9d797039be763ba8ull;
    std::vector<size_t> live, idx;
    for (size_t i = 0; i < ps.size(); ++i)
        if (!ps[i].inf && !qs[i]->inf) {
            live.push_back(i);
            idx.push_back(0);
        }
    Fq12 f = f12_one();
    for (int bit = 63; bit >= 0; --bit) {
        f = f12_sqr(f);
        for (size_t k = 0; k < live.size(); ++k) ell(f, qs[live[k]]->lines[idx[k]++], ps[live[k]].x, ps[live[k]].y);
        if ((lo >> bit) & 1)
            for (size_t k = 0; k < live.size(); ++k) ell(f, qs[live[k]]->lines[idx[k]++], ps[live[k]].x, ps[live[k]].y);
    }
    for (size_t k = 0; k < live.size(); ++k) {
        ell(f, qs[live[k]]->lines[idx[k]++], ps[live[k]].x, ps[live[k]].y);
        ell(f, qs[live[k]]->lines[idx[k]++], ps[live[k]].x, ps[live[k]].y);
    }
    return f;
}
// prepared lines of the last few G2 points seen by this thread (keyed by the point's 128-byte encoding + negation flag);
// shared ownership: an entry evicted while a caller still holds it stays alive until that caller is done
static inline std::shared_ptr<const G2Prepared> g2_prepared_cached(const uint8_t enc[128], bool negate, const G2Affine& q_as_used) {
    struct Entry {
        uint8_t key[129];
        std::shared_ptr<const G2Prepared> prep;
    };
    static thread_local std::vector<Entry> cache;
    static thread_local size_t next = 0;
    uint8_t key[129];
    memcpy(key, enc, 128);
    key[128] = negate ? 1 : 0;
    for (auto& e : cache)
        if (memcmp(e.key, key, 129) == 0) return e.prep;
    Entry e;
    memcpy(e.key, key, 129);
    e.prep = std::make_shared<const G2Prepared>(g2_prepare(q_as_used));
    std::shared_ptr<const G2Prepared> r = e.prep;
    if (cache.size() < 8) cache.push_back(std::move(e));
    else {
        cache[next] = std::move(e);
        next = (next + 1) % 8;
    }
    return r;
}

// f^((p^12 - 1) / r), exact
static inline Fq12 final_exponentiation(const Fq12& f0) {
    // easy part: (p^6 - 1)(p^2 + 1)
    Fq12 f = f12_mul(f12_conj(f0), f12_inv(f0));
    f = f12_mul(f12_frobenius(f12_frobenius(f)), f);
    // hard part (p^4 - p^2 + 1) / r = p^3 + (6x^2 + 1) p^2 + (-36x^3 - 18x^2 - 12x + 1) p + (-36x^3 - 30x^2 - 18x - 2)
    // (Scott, Benger, Charlemagne, Dominguez Perez, Kachisa 2009); inverses are conjugates in the cyclotomic subgroup
    const Fq12 fx = f12_pow_x(f), fx2 = f12_pow_x(fx), fx3 = f12_pow_x(fx2);
    const Fq12 fp = f12_frobenius(f), fp2 = f12_frobenius(fp), fp3 = f12_frobenius(fp2);
    const Fq12 y0 = f12_mul(f12_mul(fp, fp2), fp3);
    const Fq12 y1 = f12_conj(f);
    const Fq12 y2 = f12_frobenius(f12_frobenius(fx2));
    const Fq12 y3 = f12_conj(f12_frobenius(fx));
    const Fq12 y4 = f12_conj(f12_mul(fx, f12_frobenius(fx2)));
    const Fq12 y5 = f12_conj(fx2);
    const Fq12 y6 = f12_conj(f12_mul(fx3, f12_frobenius(fx3)));
    Fq12 t0 = f12_sqr(y6);
    t0 = f12_mul(t0, y4);
    t0 = f12_mul(t0, y5);
    Fq12 t1 = f12_mul(y3, y5);
    t1 = f12_mul(t1, t0);
    t0 = f12_mul(t0, y2);
    t1 = f12_sqr(t1);
    t1 = f12_mul(t1, t0);
    t1 = f12_sqr(t1);
    t0 = f12_mul(t1, y1);
    t1 = f12_mul(t1, y0);
    t0 = f12_sqr(t0);
    return f12_mul(t0, t1);
}

// C-ABI encodings: G1 affine x || y (64 B, identity = zeros); G2 affine x.c0 || x.c1 || y.c0 || y.c1 (128 B, identity =
// zeros) — halo2curves' G2Affine { x: Fq2 { c0, c1 }, y }.  Returns 0 ok, 1 non-canonical coordinate, 2 not on the curve,
// 3 G2 point outside the order-r subgroup.
static inline int load_g1(const uint8_t* b, G1Affine& p) {
    bool zero = true;
    for (int i = 0; i < 64; ++i) zero &= b[i] == 0;
    p.inf = zero;
    if (!fq_from_bytes(b, p.x) || !fq_from_bytes(b + 32, p.y)) return 1;
    return g1_on_curve(p) ? 0 : 2;
}
static inline int load_g2(const uint8_t* b, G2Affine& q) {
    bool zero = true;
    for (int i = 0; i < 128; ++i) zero &= b[i] == 0;
    q.inf = zero;
    if (!fq_from_bytes(b, q.x.c0) || !fq_from_bytes(b + 32, q.x.c1) || !fq_from_bytes(b + 64, q.y.c0) ||
        !fq_from_bytes(b + 96, q.y.c1))
        return 1;
    if (!g2_on_curve(q)) return 2;
    // the same two points ([s]_2 and [1]_2 of one ParamsKZG) come back with every batch: remember the last few that
    // passed the (0.6 ms) subgroup check
    static thread_local uint8_t seen[4][128];
    static thread_local int nseen = 0, next = 0;
    for (int i = 0; i < nseen; ++i)
        if (memcmp(seen[i], b, 128) == 0) return 0;
    if (!g2_in_subgroup(q)) return 3;
    memcpy(seen[next], b, 128);
    next = (next + 1) % 4;
    if (nseen < 4) ++nseen;
    return 0;
}
// square root in Fq2 = Fq[u]/(u^2 + 1) for p = 3 mod 4 ("complex method"): a = a0 + a1 u; with n = sqrt(a0^2 + a1^2) and
// d = (a0 +- n) / 2 a square, the root is x0 + x1 u, x0 = sqrt(d), x1 = a1 / (2 x0).  false if `a` is not a square.
static inline bool fq_sqrt(const Fq& a, Fq& out) {   // a^((p+1)/4), checked
    static const uint64_t E[4] = {0x4f082305b61f3f52ull, 0x65e05aa45a1c72a3ull, 0x6e14116da0605617ull, 0x0c19139cb84c680aull};
    const Fq r = fq_pow(a, E);
    if (!fq_eq(fq_sqr(r), a)) return false;
    out = r;
    return true;
}
static inline bool f2_sqrt(const Fq2& a, Fq2& out) {
    if (f2_is_zero(a)) {
        out = f2_zero();
        return true;
    }
    static const Fq half = fq_inv(fq_from_u64(2));
    if (fq_is_zero(a.c1)) {
        Fq r;
        if (fq_sqrt(a.c0, r)) {
            out = Fq2{r, fq_zero()};
            return true;
        }
        if (fq_sqrt(fq_neg(a.c0), r)) {   // sqrt(-1) = u
            out = Fq2{fq_zero(), r};
            return true;
        }
        return false;
    }
    Fq n;
    if (!fq_sqrt(fq_add(fq_sqr(a.c0), fq_sqr(a.c1)), n)) return false;
    for (int sign = 0; sign < 2; ++sign) {
        const Fq d = fq_mul(sign ? fq_sub(a.c0, n) : fq_add(a.c0, n), half);
        Fq x0;
        if (!fq_sqrt(d, x0) || fq_is_zero(x0)) continue;
        const Fq x1 = fq_mul(a.c1, fq_inv(fq_dbl(x0)));
        const Fq2 cand = {x0, x1};
        if (f2_eq(f2_sqr(cand), a)) {
            out = cand;
            return true;
        }
    }
    return false;
}
// G2Affine::from_bytes of halo2curves 0.2.1 (the 64-byte compressed form ParamsKZG::write stores g2 / s_g2 in; recalled
// from upstream, the crate is not vendored): x.c0 || x.c1 little-endian, bit 7 of the last byte = parity of y.c0
// (`y.to_bytes()[0] & 1`), identity = zeros.  0 ok, 1 non-canonical coordinate, 2 not on the twist.
static inline int g2_decompress(const uint8_t in[64], G2Affine& q) {
    uint8_t t[64];
    memcpy(t, in, 64);
    const int ysign = t[63] >> 7;
    t[63] &= 0x7f;
    if (!fq_from_bytes(t, q.x.c0) || !fq_from_bytes(t + 32, q.x.c1)) return 1;
    q.inf = false;
    if (f2_is_zero(q.x) && !ysign) {
        q.inf = true;
        q.y = f2_zero();
        return 0;
    }
    Fq2 y;
    if (!f2_sqrt(f2_add(f2_mul(f2_sqr(q.x), q.x), twist_b()), y)) return 2;
    uint8_t yb[32];
    fq_to_bytes(y.c0, yb);
    if ((yb[0] & 1) != ysign) y = f2_neg(y);
    q.y = y;
    return 0;
}
static inline void g2_to_bytes(const G2Affine& q, uint8_t out[128]) {
    if (q.inf) {
        memset(out, 0, 128);
        return;
    }
    fq_to_bytes(q.x.c0, out);
    fq_to_bytes(q.x.c1, out + 32);
    fq_to_bytes(q.y.c0, out + 64);
    fq_to_bytes(q.y.c1, out + 96);
}
static inline void f12_to_bytes(const Fq12& a, uint8_t* out) {   // 12 x 32 B: c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1
    const Fq2* cs[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; ++i) {
        fq_to_bytes(cs[i]->c0, out + 64 * i);
        fq_to_bytes(cs[i]->c1, out + 64 * i + 32);
    }
}

// what the C ABI's entry points (h2agg.hip) need of this namespace, as one type they are templated on
struct Api {
    typedef G1Affine G1;
    typedef G2Affine G2;
    typedef G2Prepared Prepared;
    static int load1(const uint8_t* b, G1& p) { return load_g1(b, p); }
    static int load2(const uint8_t* b, G2& q) { return load_g2(b, q); }
    static void negate(G2& q) {
        if (!q.inf) q.y = f2_neg(q.y);
    }
    static std::shared_ptr<const Prepared> prepared(const uint8_t enc[128], bool negated, const G2& q_as_used) {
        return g2_prepared_cached(enc, negated, q_as_used);
    }
    static bool check(const std::vector<G1>& ps, const std::vector<G2>& qs) {
        return f12_is_one(final_exponentiation(multi_miller_loop(ps, qs)));
    }
    static bool check_prepared(const std::vector<G1>& ps, const std::vector<const Prepared*>& qs) {
        return f12_is_one(final_exponentiation(multi_miller_loop_prepared(ps, qs)));
    }
    // the two halves of check_prepared, for a caller that runs the pairs' Miller loops on two threads: the product of the
    // loops' values is the loop of the product (each thread then pays the 64 squarings itself, and half the line work)
    typedef Fq12 Gt;
    static Gt miller_prepared(const std::vector<G1>& ps, const std::vector<const Prepared*>& qs) { return multi_miller_loop_prepared(ps, qs); }
    static bool check_product(const Gt& a, const Gt& b) { return f12_is_one(final_exponentiation(f12_mul(a, b))); }
    static void product_prepared(const std::vector<G1>& ps, const std::vector<const Prepared*>& qs, uint8_t out_gt[384]) {
        f12_to_bytes(final_exponentiation(multi_miller_loop_prepared(ps, qs)), out_gt);
    }
};

}  // namespace PAIRING_NS
}  // namespace h2agg
