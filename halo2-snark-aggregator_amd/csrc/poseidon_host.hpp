// Poseidon parameter generation on the host (C++), for the device sponge of csrc/poseidon_kernels.hpp.
//
// Stands behind `Spec::<Fr, T, RATE>::new(r_f, r_p)` of the `poseidon` crate (privacy-scaling-explorations/poseidon rev
// 0b9965fb, reference Cargo.lock:2517-2519; not vendored), which the reference instantiates through
//   PoseidonChip::new                 halo2-snark-aggregator-api/src/hash/poseidon.rs:149-165
//   PoseidonTranscriptRead::new(.., 8, 63) with T = 9, RATE = 8
//                                     halo2-snark-aggregator-circuit/src/verify_circuit.rs:127-135,154-162
// Published algorithm restated: the Poseidon paper's Grain-LFSR generator (80-bit state, taps 62 51 38 23 13 0, 160
// warm-up steps, self-shrinking output; round constants by rejection sampling of 254-bit big-endian draws, MDS = Cauchy
// matrix 1 / (x_i + y_j) from 2T draws reduced mod r), then the optimized schedule the reference's `permutation` walks
// (poseidon.rs:193-230): constants folded through M^-1, the partial rounds' matrices factored into sparse ones.
// tests/test_gpu_poseidon.py checks the device sponge against the oracle, whose generator reproduces the published
// poseidonperm_x5_254_3 / _5 vectors.
//
// This is setup code (a few milliseconds, once per process): plain 4 x 64-bit Montgomery arithmetic over Fr.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

namespace h2agg {
namespace poseidon_host {

typedef unsigned __int128 u128;

struct HFr {
    uint64_t l[4];
};
static const uint64_t R_MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t R_INV = 0xc2e1f593efffffffull;   // -r^-1 mod 2^64

static inline bool geq_mod(const uint64_t* a) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > R_MOD[i]) return true;
        if (a[i] < R_MOD[i]) return false;
    }
    return true;
}
static inline void sub_mod(uint64_t* a) {
    u128 b = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a[i] - R_MOD[i] - (uint64_t)b;
        a[i] = (uint64_t)t;
        b = (t >> 64) & 1;
    }
}
static inline HFr add(const HFr& a, const HFr& b) {
    HFr r;
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || geq_mod(r.l)) sub_mod(r.l);
    return r;
}
static inline HFr sub(const HFr& a, const HFr& b) {
    HFr r;
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a.l[i] - b.l[i] - (uint64_t)br;
        r.l[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)r.l[i] + R_MOD[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
static inline HFr mul(const HFr& a, const HFr& b) {   // Montgomery product (CIOS)
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
        const uint64_t m = t[0] * R_INV;
        c = (u128)m * R_MOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * R_MOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    HFr r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_mod(r.l)) sub_mod(r.l);
    return r;
}
struct Consts {
    HFr one, r2;
    Consts() {
        HFr x = {{1, 0, 0, 0}};
        for (int i = 0; i < 256; ++i) x = add(x, x);
        one = x;
        for (int i = 0; i < 256; ++i) x = add(x, x);
        r2 = x;
    }
};
static inline const Consts& consts() {
    static const Consts c;
    return c;
}
static inline HFr zero() { return HFr{{0, 0, 0, 0}}; }
static inline HFr one() { return consts().one; }
static inline bool is_zero(const HFr& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline HFr from_words(const uint64_t w[4]) {   // canonical integer < r -> Montgomery
    HFr t = {{w[0], w[1], w[2], w[3]}};
    return mul(t, consts().r2);
}
static inline void to_canonical(const HFr& a, uint64_t out[4]) {
    const HFr one_int = {{1, 0, 0, 0}};
    const HFr t = mul(a, one_int);
    memcpy(out, t.l, 32);
}
static inline HFr inv(const HFr& a) {   // Fermat; a != 0
    uint64_t e[4] = {R_MOD[0] - 2, R_MOD[1], R_MOD[2], R_MOD[3]};
    HFr acc = one();
    for (int i = 255; i >= 0; --i) {
        acc = mul(acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = mul(acc, a);
    }
    return acc;
}

// ---- Grain LFSR -------------------------------------------------------------------------------------------------
struct Grain {
    bool s[80];
    int head = 0;   // ring buffer: s[(head + i) % 80] is bit i of the register
    Grain(int t, int r_f, int r_p) {
        int n = 0;
        auto push = [&](uint32_t v, int bits) {
            for (int i = bits - 1; i >= 0; --i) s[n++] = (v >> i) & 1;
        };
        push(1, 2);        // prime field
        push(0, 4);        // x^alpha S-box
        push(254, 12);     // field size in bits
        push((uint32_t)t, 12);
        push((uint32_t)r_f, 10);
        push((uint32_t)r_p, 10);
        push(0x3fffffffu, 30);
        for (int i = 0; i < 160; ++i) step();
    }
    bool at(int i) const { return s[(head + i) % 80]; }
    bool step() {
        const bool b = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
        s[head] = b;               // the slot of bit 0 becomes the new bit 79
        head = (head + 1) % 80;
        return b;
    }
    bool bit() {
        for (;;) {
            const bool b1 = step(), b2 = step();
            if (b1) return b2;
        }
    }
    // 254 bits, most significant first -> 256-bit little-endian words
    void draw(uint64_t w[4]) {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (int i = 253; i >= 0; --i)
            if (bit()) w[i / 64] |= (uint64_t)1 << (i % 64);
    }
    HFr field_element() {              // rejection sampling
        uint64_t w[4];
        for (;;) {
            draw(w);
            if (!geq_mod(w)) return from_words(w);
        }
    }
    HFr field_element_without_rejection() {   // reduce mod r (a 254-bit draw is < 2r... up to 5r/4: subtract while >= r)
        uint64_t w[4];
        draw(w);
        while (geq_mod(w)) sub_mod(w);
        return from_words(w);
    }
};

// ---- small dense linear algebra over Fr ------------------------------------------------------------------------
typedef std::vector<std::vector<HFr>> Mat;
static inline Mat mat_mul(const Mat& a, const Mat& b) {
    const size_t n = a.size(), k = b.size(), p = b[0].size();
    Mat r(n, std::vector<HFr>(p, zero()));
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < p; ++j) {
            HFr acc = zero();
            for (size_t x = 0; x < k; ++x) acc = add(acc, mul(a[i][x], b[x][j]));
            r[i][j] = acc;
        }
    return r;
}
static inline std::vector<HFr> mat_vec(const Mat& m, const std::vector<HFr>& v) {
    std::vector<HFr> r(m.size(), zero());
    for (size_t i = 0; i < m.size(); ++i) {
        HFr acc = zero();
        for (size_t j = 0; j < v.size(); ++j) acc = add(acc, mul(m[i][j], v[j]));
        r[i] = acc;
    }
    return r;
}
static inline Mat mat_inv(const Mat& m) {   // Gauss-Jordan; m is invertible (Cauchy / its minors)
    const size_t n = m.size();
    Mat a(n, std::vector<HFr>(2 * n, zero()));
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = 0; j < n; ++j) a[i][j] = m[i][j];
        a[i][n + i] = one();
    }
    for (size_t c = 0; c < n; ++c) {
        size_t piv = c;
        while (piv < n && is_zero(a[piv][c])) ++piv;
        if (piv == n) return Mat();   // singular: never for these matrices
        std::swap(a[c], a[piv]);
        const HFr iv = inv(a[c][c]);
        for (size_t j = 0; j < 2 * n; ++j) a[c][j] = mul(a[c][j], iv);
        for (size_t r = 0; r < n; ++r) {
            if (r == c || is_zero(a[r][c])) continue;
            const HFr f = a[r][c];
            for (size_t j = 0; j < 2 * n; ++j) a[r][j] = sub(a[r][j], mul(f, a[c][j]));
        }
    }
    Mat out(n, std::vector<HFr>(n));
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) out[i][j] = a[i][n + j];
    return out;
}

// ---- Spec ------------------------------------------------------------------------------------------------------
struct Spec {
    int t, r_f, r_p;
    Mat start;                      // [r_f/2 + 1][t]
    std::vector<HFr> partial;       // [r_p]
    Mat end;                        // [r_f/2 - 1][t]
    Mat mds, pre_sparse;            // [t][t]
    Mat sparse_row;                 // [r_p][t]
    Mat sparse_col;                 // [r_p][t - 1]
    bool ok = false;

    Spec(int t_, int r_f_, int r_p_) : t(t_), r_f(r_f_), r_p(r_p_) {
        Grain g(t, r_f, r_p);
        Mat rc(r_f + r_p, std::vector<HFr>(t));
        for (auto& row : rc)
            for (auto& c : row) c = g.field_element();
        std::vector<HFr> xs(t), ys(t);
        for (auto& x : xs) x = g.field_element_without_rejection();
        for (auto& y : ys) y = g.field_element_without_rejection();
        mds.assign(t, std::vector<HFr>(t));
        for (int i = 0; i < t; ++i)
            for (int j = 0; j < t; ++j) {
                const HFr s = add(xs[i], ys[j]);
                if (is_zero(s)) return;
                mds[i][j] = inv(s);
            }
        const Mat mi = mat_inv(mds);
        if (mi.empty()) return;
        const int h = r_f / 2;
        start.push_back(rc[0]);
        for (int k = 1; k < h; ++k) start.push_back(mat_vec(mi, rc[k]));
        std::vector<HFr> acc = rc[h + r_p];
        partial.assign(r_p, zero());
        for (int k = r_p - 1; k >= 0; --k) {
            std::vector<HFr> tmp = mat_vec(mi, acc);
            partial[k] = tmp[0];
            tmp[0] = zero();
            for (int i = 0; i < t; ++i) acc[i] = add(tmp[i], rc[h + k][i]);
        }
        start.push_back(mat_vec(mi, acc));
        for (int k = h + r_p + 1; k < 2 * h + r_p; ++k) end.push_back(mat_vec(mi, rc[k]));
        // sparse factorisation, last partial round first (see oracle/poseidon.py for the derivation)
        Mat acc_m = mds;
        std::vector<std::vector<HFr>> rows, cols;
        for (int k = 0; k < r_p; ++k) {
            Mat hat(t - 1, std::vector<HFr>(t - 1));
            for (int i = 1; i < t; ++i)
                for (int j = 1; j < t; ++j) hat[i - 1][j - 1] = acc_m[i][j];
            const Mat hat_inv = mat_inv(hat);
            if (hat_inv.empty()) return;
            std::vector<HFr> row(t), col(t - 1);
            row[0] = acc_m[0][0];
            for (int j = 0; j < t - 1; ++j) {
                HFr s = zero();
                for (int i = 0; i < t - 1; ++i) s = add(s, mul(acc_m[0][i + 1], hat_inv[i][j]));
                row[j + 1] = s;
            }
            for (int i = 1; i < t; ++i) col[i - 1] = acc_m[i][0];
            rows.push_back(row);
            cols.push_back(col);
            Mat m_prime(t, std::vector<HFr>(t, zero()));
            m_prime[0][0] = one();
            for (int i = 1; i < t; ++i)
                for (int j = 1; j < t; ++j) m_prime[i][j] = hat[i - 1][j - 1];
            acc_m = mat_mul(m_prime, mds);
        }
        sparse_row.assign(rows.rbegin(), rows.rend());
        sparse_col.assign(cols.rbegin(), cols.rend());
        pre_sparse = acc_m;
        ok = scale_partial_rounds();
    }

    // The partial rounds with one multiplication less on the S-box lane (csrc/poseidon_kernels.hpp).  A partial round is
    //     s0' = s0^5 + c_k,   s0 <- row_0 s0' + sum_{i>=1} row_i s_i,   s_i <- s_i + col_i s0'.
    // Track w with s0 = beta_k w (beta_0 = 1) and z = w^5, and the other words without their constant parts,
    // s_i = shat_i + cum_{k,i}.  Then, with b5 = beta_k^5 and beta_{k+1} = row_0 b5,
    //     shat_i <- shat_i + A_{k,i} z            A_{k,i} = col_i b5          cum_{k+1,i} = cum_{k,i} + col_i c_k
    //     w      <- z + D_k + sum_i R_{k,i} shat_i  R_{k,i} = row_i / beta_{k+1}   D_k = c_k / b5 + sum_i R_{k,i} cum_{k,i}
    // — exact field identities; the row's multiplication by row_0 after the S-box is gone from the chain, and the lanes'
    // A z product of round k shares an issue slot with the S-box lane's first squaring of round k + 1.
    Mat ps_a, ps_r;                 // [r_p][t - 1]
    std::vector<HFr> ps_d;          // [r_p]
    std::vector<HFr> ps_fin;        // [t]: beta_{r_p}, cum_{r_p, 1..t-1}
    bool scale_partial_rounds() {
        HFr beta = one();
        std::vector<HFr> cum(t - 1, zero());
        ps_a.assign(r_p, std::vector<HFr>(t - 1));
        ps_r.assign(r_p, std::vector<HFr>(t - 1));
        ps_d.assign(r_p, zero());
        for (int k = 0; k < r_p; ++k) {
            const HFr b2 = mul(beta, beta), b4 = mul(b2, b2), b5 = mul(b4, beta);
            const HFr beta_n = mul(sparse_row[k][0], b5);
            if (is_zero(beta_n)) return false;
            const HFr ib = inv(beta_n);
            HFr d = mul(partial[k], inv(b5));
            for (int i = 0; i < t - 1; ++i) {
                ps_a[k][i] = mul(sparse_col[k][i], b5);
                ps_r[k][i] = mul(sparse_row[k][i + 1], ib);
                d = add(d, mul(ps_r[k][i], cum[i]));
            }
            ps_d[k] = d;
            for (int i = 0; i < t - 1; ++i) cum[i] = add(cum[i], mul(sparse_col[k][i], partial[k]));
            beta = beta_n;
        }
        ps_fin.assign(t, zero());
        ps_fin[0] = beta;
        for (int i = 0; i < t - 1; ++i) ps_fin[i + 1] = cum[i];
        return true;
    }
};

}  // namespace poseidon_host
}  // namespace h2agg
