"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

The BN254 optimal-ate pairing check behind the reference's accept / reject signal:

    evaluate_multiopen_proof      halo2-snark-aggregator-api/src/systems/halo2/verify.rs:733-739
        E::multi_miller_loop(&[(&left_v, &s_g2_prepared), (&right_v, &n_g2_prepared)]).final_exponentiation().is_identity()
    calc_verify_circuit_final_pair  halo2-snark-aggregator-circuit/src/verify_circuit.rs:175-199   (same, debug_assert!)
    the EVM side                  halo2-snark-aggregator-solidity/templates/verifier.sol:5-37      (precompile 0x08)

Arithmetic restated: halo2curves 0.2.1 `bn256::{G2Affine, Gt, multi_miller_loop, final_exponentiation}` (unvendored).
Only the BOOLEAN is observable in the reference, and it does not depend on Miller-loop conventions.  To stay independent
of the product's tower implementation (csrc/pairing.hpp: Fq2 -> Fq6 -> Fq12, sparse lines, cyclotomic exponentiation)
this file works in the flat polynomial basis Fq12 = Fq[w] / (w^12 - 18 w^6 + 82), maps G2 through the twist into
E(Fq12), runs the Miller loop with affine chord-and-tangent lines over Fq12 and raises to (p^12 - 1) / r with one plain
integer exponentiation.

Pins (tests/test_oracle_pairing.py, tests/golden/eip_kats.json): the G2 generator of EIP-197 (on the twist, order r),
e(G1, G2) != 1, e(G1, G2)^r = 1, bilinearity, and the public EIP-197 / go-ethereum bn256Pairing vectors.
"""
from __future__ import annotations

from .bn254 import P, R, INF, inv

ATE_LOOP_COUNT = 29793968203157093288          # 6x + 2, x = 4965661367192848881
BN_X = 4965661367192848881
assert ATE_LOOP_COUNT == 6 * BN_X + 2
assert P == 36 * BN_X ** 4 + 36 * BN_X ** 3 + 24 * BN_X ** 2 + 6 * BN_X + 1
assert R == 36 * BN_X ** 4 + 36 * BN_X ** 3 + 18 * BN_X ** 2 + 6 * BN_X + 1

# EIP-197's P2 (the generator halo2curves' G2::generator() also uses); coefficients (c0, c1) of c0 + c1*u, u^2 = -1
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


# ------------------------------------------------------------------------------------- Fq2 (pairs) — G2 group law
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % P, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


XI = (9, 1)
B2 = f2_mul((3, 0), f2_inv(XI))                 # twist: y^2 = x^3 + 3 / (9 + u)


def g2_on_curve(q) -> bool:
    if q is INF:
        return True
    x, y = q
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == (0, 0)


def g2_neg(q):
    return INF if q is INF else (q[0], ((-q[1][0]) % P, (-q[1][1]) % P))


def g2_add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return INF
        lam = f2_mul(f2_scalar(f2_mul(x1, x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(k: int, q):
    acc = INF
    for i in range(k.bit_length() - 1, -1, -1):
        acc = g2_add(acc, acc)
        if (k >> i) & 1:
            acc = g2_add(acc, q)
    return acc


# ------------------------------------------------------------------------------------- Fq12, flat: w^12 = 18 w^6 - 82
def f12(coeffs):
    return tuple(c % P for c in coeffs)


F12_ONE = f12([1] + [0] * 11)
F12_ZERO = f12([0] * 12)


def f12_add(a, b):
    return tuple((x + y) % P for x, y in zip(a, b))


def f12_sub(a, b):
    return tuple((x - y) % P for x, y in zip(a, b))


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):                 # w^k = 18 w^(k-6) - 82 w^(k-12)
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return tuple(x % P for x in t[:12])


def f12_pow(a, e: int):
    out, base = F12_ONE, a
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_mul(base, base)
        e >>= 1
    return out


def _poly_rounded_div(a, b):
    dega, degb = _deg(a), _deg(b)
    temp, o = list(a), [0] * len(a)
    for i in range(dega - degb, -1, -1):
        o[i] = (o[i] + temp[degb + i] * inv(b[degb], P)) % P
        for c in range(degb + 1):
            temp[c + i] = (temp[c + i] - o[c]) % P
    return [x % P for x in o[:_deg(o) + 1]]


def _deg(p):
    d = len(p) - 1
    while d and p[d] % P == 0:
        d -= 1
    return d


def f12_inv(a):
    """extended Euclid in Fq[w] against the modulus polynomial"""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, (-18) % P, 0, 0, 0, 0, 0, 1]
    while _deg(low):
        r = _poly_rounded_div(high, low)
        r += [0] * (13 - len(r))
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] -= lm[i] * r[j]
                new[i + j] -= low[i] * r[j]
        nm = [x % P for x in nm]
        new = [x % P for x in new]
        lm, low, hm, high = nm, new, lm, low
    iv = inv(low[0], P)
    return tuple(x * iv % P for x in lm[:12])


W = f12([0, 1] + [0] * 10)
W2, W3 = f12_mul(W, W), f12_mul(f12_mul(W, W), W)


def f12_from_f2(a):
    """c0 + c1*u with u = w^6 - 9"""
    return f12([a[0] - 9 * a[1], 0, 0, 0, 0, 0, a[1], 0, 0, 0, 0, 0])


def f12_from_tower(c):
    """product's tower element: 12 Fq coefficients ordered [c0.c0.(c0,c1), c0.c1.(..), c0.c2.(..), c1.c0.(..), ...]
    = sum over (k in w^0..1, j in v^0..2, i in u^0..1) with v = w^2 -> flat basis"""
    acc = F12_ZERO
    idx = 0
    for k in range(2):
        for j in range(3):
            t = f12_from_f2((c[idx], c[idx + 1]))
            idx += 2
            acc = f12_add(acc, f12_mul(t, f12_pow(W, 2 * j + k)))
    return acc


# ------------------------------------------------------------------------------------- curve over Fq12 + Miller loop
def twist(q):
    if q is INF:
        return INF
    x, y = q
    return (f12_mul(f12_from_f2(x), W2), f12_mul(f12_from_f2(y), W3))


def cast_g1(p):
    if p is INF:
        return INF
    return (f12([p[0]] + [0] * 11), f12([p[1]] + [0] * 11))


def e12_double(a):
    x, y = a
    lam = f12_mul(f12_mul(f12([3] + [0] * 11), f12_mul(x, x)), f12_inv(f12_add(y, y)))
    x3 = f12_sub(f12_sub(f12_mul(lam, lam), x), x)
    return (x3, f12_sub(f12_mul(lam, f12_sub(x, x3)), y))


def e12_add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        return e12_double(a) if y1 == y2 else INF
    lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    x3 = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
    return (x3, f12_sub(f12_mul(lam, f12_sub(x1, x3)), y1))


def linefunc(p1, p2, t):
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_mul(f12([3] + [0] * 11), f12_mul(x1, x1)), f12_inv(f12_add(y1, y1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def miller_loop(q12, p12):
    if q12 is INF or p12 is INF:
        return F12_ONE
    r, f = q12, F12_ONE
    for i in range(ATE_LOOP_COUNT.bit_length() - 2, -1, -1):
        f = f12_mul(f12_mul(f, f), linefunc(r, r, p12))
        r = e12_double(r)
        if (ATE_LOOP_COUNT >> i) & 1:
            f = f12_mul(f, linefunc(r, q12, p12))
            r = e12_add(r, q12)
    q1 = (f12_pow(q12[0], P), f12_pow(q12[1], P))
    nq2 = (f12_pow(q1[0], P), f12_sub(F12_ZERO, f12_pow(q1[1], P)))
    f = f12_mul(f, linefunc(r, q1, p12))
    r = e12_add(r, q1)
    f = f12_mul(f, linefunc(r, nq2, p12))
    return f


FINAL_EXP = (P ** 12 - 1) // R


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def pairing(q, p):
    """e(p, q), p in G1 (affine ints or INF), q in G2 (pairs or INF)"""
    return final_exponentiation(miller_loop(twist(q), cast_g1(p)))


def pairing_check(pairs) -> bool:
    """prod e(p_i, q_i) == 1 — EIP-197 / `multi_miller_loop(..).final_exponentiation().is_identity()`"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(twist(q), cast_g1(p)))
    return final_exponentiation(f) == F12_ONE
