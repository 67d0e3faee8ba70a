// The reference's plugin boundary for this path — the trait trio ArithCommonChip / ArithFieldChip / ArithEccChip
// (halo2-snark-aggregator-api/src/arith/{common,field,ecc}.rs) — as header-only C++ classes over the C ABI of
// include/h2agg.h, with the method names, argument meaning and failure behaviour of the Mock chips
// (halo2-snark-aggregator-api/src/mock/arith/{field,ecc}.rs).  The reference is Rust and cargo / rustc are not in this
// image, so this is the compiled-language host side above the C ABI (halo2-snark-aggregator_amd/chips.py is the same
// surface for the Python harness; halo2-snark-aggregator_amd/rust-shim/ the source-only Rust binding).
//
// Value types, as in the Mock chips:
//     AssignedValue (field) = Fr          -> Scalar: 32-byte little-endian canonical integer (`to_repr`)
//     AssignedPoint         = C::CurveExt -> Point:  96 bytes x || y || z (Jacobian), identity z = 0
//     Point (constants)     = C (affine)  -> Affine: 64 bytes x || y, identity = zeros
// The Mock chips never construct their `Error`: failure is a panic there, an exception here (ChipError::code:
// H2AGG_ERR_DIV_ZERO for `invert().unwrap()` mock/arith/field.rs:113, H2AGG_ERR_EMPTY for `acc.unwrap()`
// mock/arith/ecc.rs:128).  Single-element calls go through the batch kernels with n = 1; no arithmetic happens on the
// host, and without a HIP device the constructor throws.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "h2agg.h"

namespace h2agg_chips {

using Scalar = std::array<uint8_t, 32>;
using Affine = std::array<uint8_t, 64>;
using Point = std::array<uint8_t, 96>;

struct ChipError : std::runtime_error {
    int code;
    ChipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// MockChipCtx (mock/arith/field.rs:11-21): `point_list`, `tag`, Display
struct GpuChipCtx {
    std::vector<std::string> point_list;
    std::string tag;
    std::string display() const { return "(total points: " + std::to_string(point_list.size()) + ")"; }
};

// one context (device buffers, streams) shared by the two chips, as MockEccChip owns its MockFieldChip's world
class Gpu {
  public:
    explicit Gpu(int device = 0) {
        const int rc = h2agg_create(device, &ctx_);
        if (rc != H2AGG_OK) throw ChipError(rc, "h2agg_create failed: a HIP device is required (there is no CPU mode)");
    }
    ~Gpu() { h2agg_destroy(ctx_); }
    Gpu(const Gpu&) = delete;
    Gpu& operator=(const Gpu&) = delete;
    h2agg_ctx* ctx() const { return ctx_; }
    void check(int rc) const {
        if (rc != H2AGG_OK) throw ChipError(rc, h2agg_last_error(ctx_));
    }

  private:
    h2agg_ctx* ctx_ = nullptr;
};

inline Scalar scalar_from_u64(uint64_t v) {
    Scalar s{};
    for (int i = 0; i < 8; ++i) s[i] = (uint8_t)(v >> (8 * i));
    return s;
}

// ArithFieldChip over Fr (arith/field.rs:6-105; MockFieldChip mock/arith/field.rs:23-146)
class GpuFieldChip {
  public:
    explicit GpuFieldChip(const Gpu& g) : g_(g) {}
    // ---- ArithCommonChip (arith/common.rs:3-42)
    Scalar add(GpuChipCtx&, const Scalar& a, const Scalar& b) const { return op(H2AGG_OP_ADD, a, &b); }   // mock field.rs:39-46
    Scalar sub(GpuChipCtx&, const Scalar& a, const Scalar& b) const { return op(H2AGG_OP_SUB, a, &b); }   // :48-55
    Scalar assign_zero(GpuChipCtx&) const { return Scalar{}; }
    Scalar assign_one(GpuChipCtx&) const { return scalar_from_u64(1); }
    Scalar assign_const(GpuChipCtx&, const Scalar& c) const { return c; }
    Scalar assign_var(GpuChipCtx&, const Scalar& v) const { return v; }
    Scalar to_value(const Scalar& v) const { return v; }
    Scalar normalize(GpuChipCtx&, const Scalar& v) const { return v; }
    // ---- ArithFieldChip
    Scalar mul(GpuChipCtx&, const Scalar& a, const Scalar& b) const { return op(H2AGG_OP_MUL, a, &b); }   // :98-105
    Scalar div(GpuChipCtx&, const Scalar& a, const Scalar& b) const { return op(H2AGG_OP_DIV, a, &b); }   // :107-114, b = 0 panics
    Scalar square(GpuChipCtx&, const Scalar& a) const { return op(H2AGG_OP_SQR, a, nullptr); }            // :116-122
    // acc = b; acc += x * coeff   (mock/arith/field.rs:124-135), one launch
    Scalar sum_with_coeff_and_constant(GpuChipCtx&, const std::vector<std::pair<Scalar, Scalar>>& a_with_coeff,
                                       const Scalar& b) const {
        if (a_with_coeff.empty()) return b;
        std::vector<uint8_t> x(32 * a_with_coeff.size()), c(32 * a_with_coeff.size());
        for (size_t i = 0; i < a_with_coeff.size(); ++i) {
            std::memcpy(&x[32 * i], a_with_coeff[i].first.data(), 32);
            std::memcpy(&c[32 * i], a_with_coeff[i].second.data(), 32);
        }
        Scalar out;
        g_.check(h2agg_fr_sum_with_coeff_and_constant(g_.ctx(), x.data(), c.data(), a_with_coeff.size(), b.data(), out.data()));
        return out;
    }
    Scalar sum_with_constant(GpuChipCtx& ctx, const std::vector<Scalar>& a, const Scalar& b) const {   // arith/field.rs:37-48
        std::vector<std::pair<Scalar, Scalar>> v;
        for (const Scalar& x : a) v.emplace_back(x, scalar_from_u64(1));
        return sum_with_coeff_and_constant(ctx, v, b);
    }
    Scalar mul_add_constant(GpuChipCtx& ctx, const Scalar& a, const Scalar& b, const Scalar& c) const {   // mock field.rs:137-145
        return add(ctx, mul(ctx, a, b), c);
    }
    Scalar mul_add(GpuChipCtx& ctx, const Scalar& a, const Scalar& b, const Scalar& c) const {            // arith/field.rs:57-66
        return add(ctx, mul(ctx, a, b), c);
    }
    // Horner: acc = a[0]; acc = acc * b + a[i]   (arith/field.rs:68-81)
    Scalar mul_add_accumulate(GpuChipCtx& ctx, const std::vector<Scalar>& a, const Scalar& b) const {
        if (a.empty()) return assign_zero(ctx);
        std::vector<uint8_t> v(32 * a.size());
        for (size_t i = 0; i < a.size(); ++i) std::memcpy(&v[32 * i], a[i].data(), 32);
        Scalar out;
        g_.check(h2agg_fr_mul_add_accumulate(g_.ctx(), v.data(), a.size(), b.data(), out.data()));
        return out;
    }
    Scalar pow_constant(GpuChipCtx&, const Scalar& base, uint32_t exponent) const {                       // arith/field.rs:83-104 (asserts >= 1)
        Scalar out;
        g_.check(h2agg_fr_batch_pow_constant(g_.ctx(), base.data(), 1, exponent, out.data()));
        return out;
    }

  private:
    Scalar op(int which, const Scalar& a, const Scalar* b) const {
        Scalar out;
        g_.check(h2agg_fr_batch_op(g_.ctx(), which, a.data(), b ? b->data() : nullptr, 1, out.data()));
        return out;
    }
    const Gpu& g_;
};

// ArithEccChip over BN254 G1 (arith/ecc.rs:5-61; MockEccChip mock/arith/ecc.rs:8-130)
class GpuEccChip {
  public:
    explicit GpuEccChip(const Gpu& g) : g_(g) {}
    Point add(GpuChipCtx&, const Point& a, const Point& b) const { return addsub(a, b, 0); }   // mock ecc.rs:30-37
    Point sub(GpuChipCtx&, const Point& a, const Point& b) const { return addsub(a, b, 1); }   // :39-46
    Point assign_zero(GpuChipCtx&) const {                                                     // :48-50
        Point p{};
        p[32] = 1;
        return p;
    }
    Point assign_one(GpuChipCtx& ctx) const {                                                  // :52-54: the generator (1, 2)
        Affine g{};
        g[0] = 1;
        g[32] = 2;
        return assign_const(ctx, g);
    }
    Point assign_const(GpuChipCtx& ctx, const Affine& c) const {                               // :56-62: to_curve
        bool ident = true;
        for (uint8_t b : c) ident = ident && b == 0;
        if (ident) return assign_zero(ctx);
        Point p{};
        std::memcpy(p.data(), c.data(), 64);
        p[64] = 1;
        return p;
    }
    Point assign_var(GpuChipCtx& ctx, const Affine& v) const { return assign_const(ctx, v); }
    Affine to_value(const Point& v) const {                                                    // :64-66: to_affine
        Affine out;
        g_.check(h2agg_g1_batch_to_affine(g_.ctx(), v.data(), 1, out.data()));
        return out;
    }
    Point normalize(GpuChipCtx&, const Point& v) const { return v; }                           // :68-74: the identity function
    Point scalar_mul(GpuChipCtx&, const Scalar& lhs, const Point& rhs) const {                 // :88-95: rhs * lhs
        const Affine a = to_value(rhs);
        Point out;
        g_.check(h2agg_g1_batch_scalar_mul(g_.ctx(), a.data(), lhs.data(), 1, out.data()));
        return out;
    }
    Point scalar_mul_constant(GpuChipCtx&, const Scalar& lhs, const Affine& rhs) const {       // :97-104
        Point out;
        g_.check(h2agg_g1_batch_scalar_mul(g_.ctx(), rhs.data(), lhs.data(), 1, out.data()));
        return out;
    }
    // mock/arith/ecc.rs:106-129: records the points in ctx.point_list (their count is what Display shows), then
    // sum_i scalars[i] * points[i] — one Pippenger MSM over the projective points as they are (h2agg_g1_msm_jac).
    // Zero pairs: the reference panics (`acc.unwrap()`), here ChipError with H2AGG_ERR_EMPTY.
    Point multi_exp(GpuChipCtx& ctx, const std::vector<Point>& points, const std::vector<Scalar>& scalars) const {
        ctx.point_list.assign(points.size(), std::string());   // (the strings are Debug output of the reference's type: not reproduced)
        const size_t n = points.size() < scalars.size() ? points.size() : scalars.size();
        std::vector<uint8_t> pb(96 * n), sb(32 * n);
        for (size_t i = 0; i < n; ++i) {
            std::memcpy(&pb[96 * i], points[i].data(), 96);
            std::memcpy(&sb[32 * i], scalars[i].data(), 32);
        }
        Point out;
        g_.check(h2agg_g1_msm_jac(g_.ctx(), pb.data(), sb.data(), n, out.data()));
        return out;
    }

  private:
    Point addsub(const Point& a, const Point& b, int subtract) const {
        Point out;
        g_.check(h2agg_g1_batch_add(g_.ctx(), a.data(), b.data(), 1, subtract, out.data()));
        return out;
    }
    const Gpu& g_;
};

// ---- EvaluationQuerySchema (halo2-snark-aggregator-api/src/systems/halo2/evaluation.rs:15-330) --------------------
// The AST with the reference's constructors and operators: commit!(cq) / eval!(cq) / scalar!(s) (:41-60), `impl Add` /
// `impl Mul` (:62-84), EvaluationQuery::new (:100-118), estimate (:295-330) and eval (:172-203).  Nodes live in an arena
// owned by a SchemaArena (one per verification, like the reference's short-lived trees); the scalar bookkeeping of
// eval_prepare runs as a tape on the device and feeds the multi_exp directly.
class SchemaArena;
class EvaluationQuerySchema {
  public:
    EvaluationQuerySchema operator+(const EvaluationQuerySchema& r) const;   // evaluation.rs:62-72
    EvaluationQuerySchema operator*(const EvaluationQuerySchema& r) const;   // :74-84
    size_t estimate() const;                                                 // estimate(None), :295-330
    // eval(): (point, Some(scalar) / None).  Errors as the reference's assert! / panic (H2AGG_ERR_INVALID / _EMPTY).
    struct Evaluated {
        Point point;
        bool has_scalar;
        Scalar scalar;
    };
    Evaluated eval() const;
    uint32_t node() const { return id_; }

  private:
    friend class SchemaArena;
    EvaluationQuerySchema(SchemaArena* a, uint32_t id) : a_(a), id_(id) {}
    SchemaArena* a_;
    uint32_t id_;
};

class SchemaArena {
  public:
    explicit SchemaArena(const Gpu& g) : g_(g) { g_.check(h2agg_schema_create(g_.ctx(), &s_)); }
    ~SchemaArena() { h2agg_schema_destroy(s_); }
    SchemaArena(const SchemaArena&) = delete;
    SchemaArena& operator=(const SchemaArena&) = delete;
    // commit!(x): CommitQuerySchema { key, commitment: Some(point), eval: .. }
    EvaluationQuerySchema commit(const std::string& key, const Affine& point) {
        uint32_t id;
        g_.check(h2agg_schema_node_commitment(s_, key.c_str(), point.data(), &id));
        return EvaluationQuerySchema(this, id);
    }
    EvaluationQuerySchema eval(const Scalar& e) {      // eval!(x)
        uint32_t id;
        g_.check(h2agg_schema_node_eval(s_, e.data(), &id));
        return EvaluationQuerySchema(this, id);
    }
    EvaluationQuerySchema scalar(const Scalar& v) {    // scalar!(x)
        uint32_t id;
        g_.check(h2agg_schema_node_scalar(s_, v.data(), &id));
        return EvaluationQuerySchema(this, id);
    }
    // EvaluationQuery::new(rotation, point, key, commitment, eval).s = commit!(cq) + eval!(cq)   (:100-118)
    EvaluationQuerySchema query(const std::string& key, const Affine& commitment, const Scalar& e) {
        const char* k = key.c_str();
        uint32_t id;
        g_.check(h2agg_schema_evaluation_queries(s_, 1, &k, commitment.data(), e.data(), &id));
        return EvaluationQuerySchema(this, id);
    }
    // names of the last eval (evaluation.rs:183)
    std::vector<std::string> names() const {
        std::vector<std::string> out;
        for (size_t i = 0; i < h2agg_schema_name_count(s_); ++i) out.emplace_back(h2agg_schema_name(s_, i));
        return out;
    }
    h2agg_schema* raw() const { return s_; }
    const Gpu& gpu() const { return g_; }

  private:
    const Gpu& g_;
    h2agg_schema* s_ = nullptr;
};

inline EvaluationQuerySchema EvaluationQuerySchema::operator+(const EvaluationQuerySchema& r) const {
    uint32_t id;
    a_->gpu().check(h2agg_schema_node_add(a_->raw(), id_, r.id_, &id));
    return EvaluationQuerySchema(a_, id);
}
inline EvaluationQuerySchema EvaluationQuerySchema::operator*(const EvaluationQuerySchema& r) const {
    uint32_t id;
    a_->gpu().check(h2agg_schema_node_mul(a_->raw(), id_, r.id_, &id));
    return EvaluationQuerySchema(a_, id);
}
inline size_t EvaluationQuerySchema::estimate() const {
    size_t n = 0;
    a_->gpu().check(h2agg_schema_estimate(a_->raw(), id_, &n));
    return n;
}
inline EvaluationQuerySchema::Evaluated EvaluationQuerySchema::eval() const {
    Evaluated e{};
    int has = 0;
    a_->gpu().check(h2agg_schema_eval(a_->raw(), id_, e.point.data(), &has, e.scalar.data()));
    e.has_scalar = has != 0;
    return e;
}

}  // namespace h2agg_chips
