// Per-particle / per-node arithmetic of the MPM substep, shared by every HIP kernel.
//
// Everything here is a pure function of registers (plus read-only SDF tables), templated on the
// scalar type R (float or double), and usable from host code so that tests can exercise the
// device math on the CPU (tests/harness/math_harness.cpp).  It is NOT a CPU fallback: the
// product only ever calls these from HIP kernels.
//
// Reference semantics being restated (softmac/engine/...):
//   B-spline weights              mpm_simulator.py:215-217
//   constitutive update + stress  mpm_simulator.py:219-250
//   SVD adjoint                   mpm_simulator.py:140-157, 184-192
//   quaternion helpers            primitive/primitive_utils.py:3-46
//   SDF / normal lookup           primitive/mesh.py:45-113, primitive_base.py:53-70
//   forecast contact              primitive/primitive_base.py:139-181
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SMAC_HD __host__ __device__ __forceinline__
#else
#define SMAC_HD inline
#endif

// SMAC_PRECISE=1: the SVD, the constitutive update and its adjoint keep their operation order and exact reciprocals inside the
// -ffast-math build (`#pragma clang fp`; `#pragma float_control` is not supported on amdgcn).  Measured (tools/prec_probe.py):
// it only matters for particles inside the reference's backward_svd clamp, whose gradient is ill-conditioned in any case
// (profiles/HISTORY.md 3), and costs 11 % more instructions in k_p2g_grad - off by default.
#if defined(__clang__) && defined(SMAC_PRECISE) && SMAC_PRECISE
#define SMAC_PRECISE_FP _Pragma("clang fp reassociate(off) reciprocal(off)")
#else
#define SMAC_PRECISE_FP
#endif

namespace smac {

enum : int { MODEL_COROTATED = 0, MODEL_NEOHOOKEAN = 1 };
enum : int { MAT_PLASTIC = 0, MAT_ELASTIC = 1, MAT_LIQUID = 2 };
enum : int { CONTACT_GRID = 0, CONTACT_PARTICLE = 1, CONTACT_MIXED = 2 };

constexpr int MAX_PRIMS = 4;

// ------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------
template <class R> SMAC_HD R rsqrt_(R x) { return R(1) / std::sqrt(x); }
template <class R> SMAC_HD R min_(R a, R b) { return a < b ? a : b; }
template <class R> SMAC_HD R max_(R a, R b) { return a > b ? a : b; }

// ------------------------------------------------------------------------------------------
// forward-mode dual number: used for the contact adjoint (19 inputs -> 9 outputs per primitive,
// evaluated only for the few particles inside the 5e-3 contact band).  The derivative follows
// the branch the value takes, integer casts drop it and comparison flags are constants: the
// same conventions as Taichi's reverse mode.
// ------------------------------------------------------------------------------------------
template <class R> struct Dual {
    R v, d;
    SMAC_HD Dual() : v(0), d(0) {}
    SMAC_HD Dual(R v_) : v(v_), d(0) {}
    SMAC_HD Dual(R v_, R d_) : v(v_), d(d_) {}
};
template <class R> SMAC_HD Dual<R> operator+(Dual<R> a, Dual<R> b) { return {a.v + b.v, a.d + b.d}; }
template <class R> SMAC_HD Dual<R> operator-(Dual<R> a, Dual<R> b) { return {a.v - b.v, a.d - b.d}; }
template <class R> SMAC_HD Dual<R> operator-(Dual<R> a) { return {-a.v, -a.d}; }
template <class R> SMAC_HD Dual<R> operator*(Dual<R> a, Dual<R> b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
template <class R> SMAC_HD Dual<R> operator/(Dual<R> a, Dual<R> b) {
    R inv = R(1) / b.v;
    R q = a.v * inv;
    return {q, (a.d - q * b.d) * inv};
}
template <class R> SMAC_HD Dual<R> operator+(Dual<R> a, R b) { return {a.v + b, a.d}; }
template <class R> SMAC_HD Dual<R> operator+(R a, Dual<R> b) { return {a + b.v, b.d}; }
template <class R> SMAC_HD Dual<R> operator-(Dual<R> a, R b) { return {a.v - b, a.d}; }
template <class R> SMAC_HD Dual<R> operator-(R a, Dual<R> b) { return {a - b.v, -b.d}; }
template <class R> SMAC_HD Dual<R> operator*(Dual<R> a, R b) { return {a.v * b, a.d * b}; }
template <class R> SMAC_HD Dual<R> operator*(R a, Dual<R> b) { return {a * b.v, a * b.d}; }
template <class R> SMAC_HD Dual<R> operator/(Dual<R> a, R b) { return {a.v / b, a.d / b}; }
template <class R> SMAC_HD Dual<R> operator/(R a, Dual<R> b) { return Dual<R>(a) / b; }

template <class R> SMAC_HD R val(R a) { return a; }
template <class R> SMAC_HD R val(Dual<R> a) { return a.v; }
SMAC_HD float sqrt_(float a) { return std::sqrt(a); }
SMAC_HD double sqrt_(double a) { return std::sqrt(a); }
SMAC_HD float exp_(float a) { return std::exp(a); }
SMAC_HD double exp_(double a) { return std::exp(a); }
template <class R> SMAC_HD Dual<R> sqrt_(Dual<R> a) { R s = std::sqrt(a.v); return {s, a.d / (R(2) * s)}; }
template <class R> SMAC_HD Dual<R> exp_(Dual<R> a) { R e = std::exp(a.v); return {e, e * a.d}; }
// min/max against a constant: derivative of the selected operand (ties -> the variable)
template <class S, class R> SMAC_HD S maxc(S a, R c) { return val(a) >= c ? a : S(c); }
template <class S, class R> SMAC_HD S minc(S a, R c) { return val(a) <= c ? a : S(c); }

template <class S> struct scalar_of { using type = S; };
template <class R> struct scalar_of<Dual<R>> { using type = R; };

// ------------------------------------------------------------------------------------------
// small vector helpers (generic scalar S)
// ------------------------------------------------------------------------------------------
template <class S> SMAC_HD S dot3(const S* a, const S* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class S> SMAC_HD void cross3(const S* a, const S* b, S* o) {
    S x = a[1] * b[2] - a[2] * b[1];
    S y = a[2] * b[0] - a[0] * b[2];
    S z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

// qrot, primitive_utils.py:7-13 (rot is NOT normalised here, as in the reference)
template <class S> SMAC_HD void qrot(const S* rot, const S* v, S* o) {
    using R = typename scalar_of<S>::type;
    S uv[3], uuv[3];
    cross3(rot + 1, v, uv);
    cross3(rot + 1, uv, uuv);
    S x = v[0] + R(2) * (rot[0] * uv[0] + uuv[0]);
    S y = v[1] + R(2) * (rot[0] * uv[1] + uuv[1]);
    S z = v[2] + R(2) * (rot[0] * uv[2] + uuv[2]);
    o[0] = x; o[1] = y; o[2] = z;
}

// inv_trans, primitive_utils.py:42-46
template <class S> SMAC_HD void inv_trans(const S* pos, const S* position, const S* rotation, S* o) {
    S iq[4] = {rotation[0], -rotation[1], -rotation[2], -rotation[3]};
    S n = sqrt_(iq[0] * iq[0] + iq[1] * iq[1] + iq[2] * iq[2] + iq[3] * iq[3]);
    iq[0] = iq[0] / n; iq[1] = iq[1] / n; iq[2] = iq[2] / n; iq[3] = iq[3] / n;
    S d[3] = {pos[0] - position[0], pos[1] - position[1], pos[2] - position[2]};
    qrot(iq, d, o);
}

// qmul, primitive_utils.py:19-27 ; w2quat :29-40 ; forward_kinematics primitive_base.py:280-283
template <class S> SMAC_HD void qmul(const S* q, const S* r, S* o) {
    // terms[a][b] = r[a]*q[b]
    S w = r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3];
    S x = r[0] * q[1] + r[1] * q[0] - r[2] * q[3] + r[3] * q[2];
    S y = r[0] * q[2] + r[1] * q[3] + r[2] * q[0] - r[3] * q[1];
    S z = r[0] * q[3] - r[1] * q[2] + r[2] * q[1] + r[3] * q[0];
    S n = sqrt_(w * w + x * x + y * y + z * z);
    o[0] = w / n; o[1] = x / n; o[2] = y / n; o[3] = z / n;
}
SMAC_HD float sin_(float a) { return std::sin(a); }
SMAC_HD double sin_(double a) { return std::sin(a); }
SMAC_HD float cos_(float a) { return std::cos(a); }
SMAC_HD double cos_(double a) { return std::cos(a); }
template <class R> SMAC_HD Dual<R> sin_(Dual<R> a) { return {std::sin(a.v), std::cos(a.v) * a.d}; }
template <class R> SMAC_HD Dual<R> cos_(Dual<R> a) { return {std::cos(a.v), -std::sin(a.v) * a.d}; }
template <class S> SMAC_HD void w2quat(const S* aa, S* o) {
    using R = typename scalar_of<S>::type;
    S w = sqrt_(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2] + R(1e-12));
    S s = sin_(w / R(2));
    o[0] = cos_(w / R(2));
    o[1] = (aa[0] / w) * s; o[2] = (aa[1] / w) * s; o[3] = (aa[2] / w) * s;
}
// state13 = pos3 quat4 v3 w3 ; out7 = pos3 quat4 of the next frame
template <class S, class R> SMAC_HD void forward_kinematics(const S* s13, R dt, S* out7) {
    out7[0] = s13[0] + s13[7] * dt; out7[1] = s13[1] + s13[8] * dt; out7[2] = s13[2] + s13[9] * dt;
    S aa[3] = {s13[10] * dt, s13[11] * dt, s13[12] * dt};
    S q[4];
    w2quat(aa, q);
    qmul(q, s13 + 3, out7 + 3);
}

// ------------------------------------------------------------------------------------------
// rigid primitive: constant table part + per-frame state (13 scalars: pos3 quat4 v3 w3)
// ------------------------------------------------------------------------------------------
template <class R> struct PrimTable {
    const R* sdf;      // (rx,ry,rz)
    const R* normal;   // (rx,ry,rz,3)
    int res[3];
    R lower[3], upper[3];
    R inv_dx;
    R friction, softness;
    int contact;       // primitives_contact[i]
};

// 8-tap trilinear weights/indices of mesh.py:55-65 / :99-109.  Returns false outside the box.
template <class S, class R>
SMAC_HD bool sdf_cell(const PrimTable<R>& T, const S* local, int* base, S* fx) {
    bool in_box = true;
    for (int i = 0; i < 3; ++i)
        if (val(local[i]) < T.lower[i] || val(local[i]) >= T.upper[i]) in_box = false;   // mesh.py:50-52
    if (!in_box) return false;
    for (int i = 0; i < 3; ++i) {
        S p = (local[i] - T.lower[i]) * T.inv_dx;
        int b = (int)val(p);
        if (b > T.res[i] - 2) b = T.res[i] - 2;   // guards the rounding case pos == res-1 (reference reads out of range there)
        base[i] = b;
        fx[i] = p - R(b);
    }
    return true;
}

// Primitive.sdf, primitive_base.py:53-56 + Mesh._sdf mesh.py:45-68
template <class S, class R>
SMAC_HD S prim_sdf(const PrimTable<R>& T, const S* st13, const S* pos) {
    S local[3];
    inv_trans(pos, st13, st13 + 3, local);
    int b[3]; S fx[3];
    if (!sdf_cell(T, local, b, fx)) return S(R(1e10));
    S out = S(R(0));
    const int sy = T.res[2], sx = T.res[1] * T.res[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                S w = (i ? fx[0] : R(1) - fx[0]) * (j ? fx[1] : R(1) - fx[1]) * (k ? fx[2] : R(1) - fx[2]);
                out = out + w * T.sdf[(b[0] + i) * sx + (b[1] + j) * sy + (b[2] + k)];
            }
    return out;
}

// Primitive.normal, primitive_base.py:58-61 + Mesh._normal mesh.py:90-113
template <class S, class R>
SMAC_HD void prim_normal(const PrimTable<R>& T, const S* st13, const S* pos, S* n_out) {
    S local[3];
    inv_trans(pos, st13, st13 + 3, local);
    int b[3]; S fx[3];
    S n[3] = {S(R(0)), S(R(1)), S(R(0))};
    if (sdf_cell(T, local, b, fx)) {
        n[1] = S(R(0));
        const int sy = T.res[2], sx = T.res[1] * T.res[2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    S w = (i ? fx[0] : R(1) - fx[0]) * (j ? fx[1] : R(1) - fx[1]) * (k ? fx[2] : R(1) - fx[2]);
                    const R* t = T.normal + 3 * ((b[0] + i) * sx + (b[1] + j) * sy + (b[2] + k));
                    n[0] = n[0] + w * t[0]; n[1] = n[1] + w * t[1]; n[2] = n[2] + w * t[2];
                }
        S l = sqrt_(dot3(n, n));                       // .normalized(), mesh.py:110
        n[0] = n[0] / l; n[1] = n[1] / l; n[2] = n[2] / l;
    }
    qrot(st13 + 3, n, n_out);
}

// Primitive.sdf AND Primitive.normal at the same point in one go: one inverse transform, one cell lookup, and the 8 + 24 table taps issued together
// (the forecast-contact chain is a sequence of dependent memory round trips - sdf(x), normal(x), sdf(x'), normal(x') - for a handful of particles:
// fetching the normal's taps with the distance's halves it; same arithmetic as prim_sdf / prim_normal, value for value)
template <class S, class R>
SMAC_HD S prim_sdf_normal(const PrimTable<R>& T, const S* st13, const S* pos, S* n_out) {
    S local[3];
    inv_trans(pos, st13, st13 + 3, local);
    int b[3]; S fx[3];
    S n[3] = {S(R(0)), S(R(1)), S(R(0))};
    S out = S(R(1e10));
    if (sdf_cell(T, local, b, fx)) {
        n[1] = S(R(0));
        out = S(R(0));
        const int sy = T.res[2], sx = T.res[1] * T.res[2];
        R ts[8], tn[24];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    const int c = (b[0] + i) * sx + (b[1] + j) * sy + (b[2] + k), q = 4 * i + 2 * j + k;
                    ts[q] = T.sdf[c];
                    tn[3 * q] = T.normal[3 * c]; tn[3 * q + 1] = T.normal[3 * c + 1]; tn[3 * q + 2] = T.normal[3 * c + 2];
                }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    const int q = 4 * i + 2 * j + k;
                    S w = (i ? fx[0] : R(1) - fx[0]) * (j ? fx[1] : R(1) - fx[1]) * (k ? fx[2] : R(1) - fx[2]);
                    out = out + w * ts[q];
                    n[0] = n[0] + w * tn[3 * q]; n[1] = n[1] + w * tn[3 * q + 1]; n[2] = n[2] + w * tn[3 * q + 2];
                }
        S l = sqrt_(dot3(n, n));                       // .normalized(), mesh.py:110
        n[0] = n[0] / l; n[1] = n[1] / l; n[2] = n[2] / l;
    }
    qrot(st13 + 3, n, n_out);
    return out;
}

// collider_v, primitive_base.py:63-70
template <class S> SMAC_HD void collider_v(const S* st13, const S* r, S* o) {
    const S* rot = st13 + 3;
    S n = sqrt_(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2] + rot[3] * rot[3]);
    S q[4] = {rot[0] / n, rot[1] / n, rot[2] / n, rot[3] / n};
    S iq[4] = {q[0], -q[1], -q[2], -q[3]};
    S rl[3], wl[3], vl[3];
    qrot(iq, r, rl);
    cross3(st13 + 10, rl, wl);
    vl[0] = st13[7] + wl[0]; vl[1] = st13[8] + wl[1]; vl[2] = st13[9] + wl[2];
    qrot(q, vl, o);
}

// collide_mixed, primitive_base.py:139-181.
// Returns true when the particle is inside the contact band (dist <= 5e-3); then v_io is replaced
// by the target velocity and ext6 receives this particle's wrench contribution (b_f, b_t).
template <class S, class R>
SMAC_HD bool collide_mixed(const PrimTable<R>& T, const S* st13, const S* p_pos, S* v_io,
                           R p_mass, R dt, R life, S* ext6) {
    S D[3], r[3], cv[3], in[3];
    S dist = prim_sdf_normal(T, st13, p_pos, D);                       // :141, :145 (the normal's taps travel with the distance's)
    if (!(val(dist) <= R(5e-3))) return false;                         // :142-143
    S p_v_in[3] = {v_io[0], v_io[1], v_io[2]};
    S p_v[3] = {v_io[0], v_io[1], v_io[2]};
    for (int i = 0; i < 3; ++i) r[i] = p_pos[i] - st13[i];             // :146
    collider_v(st13, r, cv);                                           // :147
    for (int i = 0; i < 3; ++i) in[i] = p_v[i] - cv[i];                // :149
    S nc = dot3(in, D);                                                // :150
    if (val(nc) < R(0)) {                                              // :152
        S t[3] = {in[0] - nc * D[0], in[1] - nc * D[1], in[2] - nc * D[2]};   // :153
        S tt = dot3(t, t);
        S tn = sqrt_(tt + R(1e-8));                                    // length(), primitive_utils.py:5
        S scale = maxc(tn + nc * T.friction, R(0)) / tn;               // :155
        R flag = (std::sqrt(val(tt)) > R(1e-30)) ? R(1) : R(0);        // :156 (nc < 0 holds here)
        for (int i = 0; i < 3; ++i) t[i] = (t[i] * scale) * flag + t[i] * (R(1) - flag);   // :157
        if (val(dist) > R(0)) {                                        // :161-163
            S infl = minc(exp_(-dist * T.softness), R(1));
            for (int i = 0; i < 3; ++i) p_v[i] = cv[i] + in[i] * (R(1) - infl) + t[i] * infl;
        } else {
            for (int i = 0; i < 3; ++i) p_v[i] = cv[i] + t[i];         // :159
        }
    }
    S xn[3] = {p_v[0] * dt + p_pos[0], p_v[1] * dt + p_pos[1], p_v[2] * dt + p_pos[2]};   // :166
    S n2[3];
    S sdf2 = prim_sdf_normal(T, st13, xn, n2);                         // :167, :169
    if (val(sdf2) < R(0)) {                                            // :168-170
        S k = (sdf2 / dt) * life;
        for (int i = 0; i < 3; ++i) p_v[i] = p_v[i] - k * n2[i];
    }
    S bf[3], bt[3];
    for (int i = 0; i < 3; ++i) bf[i] = (p_v_in[i] - p_v[i]) * (p_mass * (R(1) / dt));   // :173
    cross3(r, bf, bt);                                                 // :174
    for (int i = 0; i < 3; ++i) { ext6[i] = bf[i]; ext6[3 + i] = bt[i]; v_io[i] = p_v[i]; }
    return true;
}

// ------------------------------------------------------------------------------------------
// collide_mixed in TWO arithmetic widths (round 5; float32 storage mode only).
//
// The all-f64 chain above is ~2,000 dependent f64 instructions for each of the few thousand particles inside a contact band, its dual-number
// form ~4,500: 40 us of latency per substep pair on an otherwise idle chip (profiles/r04_ag_contact_probe.txt).  What NEEDS f64 is what the
// push-out divides by dt: the signed distance.  v -= (sdf(x') / dt) n life turns an absolute error e of the distance into e / dt of velocity, so
// the north-star's 1e-5 of a 0.3 m/s field at dt = 1e-4 allows e = 3e-10 m - a float position (3e-8 at 0.5) or a float cell fraction (6e-8 of a
// 2.5 ... 7.5 mm table cell = 1.5e-10 ... 4.5e-10) does not give that.  So S64 (double, or Dual<double> in the adjoint) carries exactly:
//   the particle position relative to the primitive, its rotation into the body frame (inv_trans, primitive_utils.py:42-46), the cell
//   fraction (mesh.py:55-57) and the 8-tap interpolation of the f64 distance table (:58-65); the forecast position x' = x + v dt (:166);
//   the product sdf(x') life / dt (:169).
// Everything else - the 24 normal taps and their normalisation (mesh.py:99-110), collider_v (:63-70), the relative velocity, Coulomb friction,
// the softness blend exp(-666 d) (:149-163), the wrench (:173-179) - is S32 (float / Dual<float>): relative errors of 1e-7 on quantities of the
// size of the velocity itself, with native rcp / rsq / exp instead of f64 Newton sequences.  ~160 f64 + ~400 f32 operations.
// Same branches, same order of operations, same derivative conventions as collide_mixed.
// ------------------------------------------------------------------------------------------
SMAC_HD float narrow_(double a) { return (float)a; }
SMAC_HD Dual<float> narrow_(Dual<double> a) { return Dual<float>((float)a.v, (float)a.d); }
SMAC_HD double widen_(float a) { return (double)a; }
SMAC_HD Dual<double> widen_(Dual<float> a) { return Dual<double>((double)a.v, (double)a.d); }

// distance (S64) and world-frame normal (S32) at `pos`.  `iq`: the NORMALISED inverse quaternion of the pose (computed once per primitive: inv_trans
// normalises it on every call, to the same value), `rot32`: the pose quaternion as it stands (primitive_base.py:61 rotates the normal back with it).
template <class S64, class S32>
SMAC_HD S64 prim_sdf_normal_hybrid(const PrimTable<double>& T64, const PrimTable<float>& T32, const S64* position, const S64* iq, const S32* rot32,
                                   const S64* pos, S32* n_out) {
    S64 d[3] = {pos[0] - position[0], pos[1] - position[1], pos[2] - position[2]};
    S64 local[3];
    qrot(iq, d, local);
    int b[3]; S64 fx[3];
    S32 n[3] = {S32(0.f), S32(1.f), S32(0.f)};
    S64 out = S64(1e10);
    if (sdf_cell(T64, local, b, fx)) {
        n[1] = S32(0.f);
        out = S64(0.0);
        const int sy = T64.res[2], sx = T64.res[1] * T64.res[2];
        double ts[8]; float tn[24];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    const int c = (b[0] + i) * sx + (b[1] + j) * sy + (b[2] + k), q = 4 * i + 2 * j + k;
                    ts[q] = T64.sdf[c];
                    tn[3 * q] = T32.normal[3 * c]; tn[3 * q + 1] = T32.normal[3 * c + 1]; tn[3 * q + 2] = T32.normal[3 * c + 2];
                }
        const S32 f32[3] = {narrow_(fx[0]), narrow_(fx[1]), narrow_(fx[2])};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    const int q = 4 * i + 2 * j + k;
                    S64 w = (i ? fx[0] : 1.0 - fx[0]) * (j ? fx[1] : 1.0 - fx[1]) * (k ? fx[2] : 1.0 - fx[2]);
                    out = out + w * ts[q];
                    S32 w32 = (i ? f32[0] : 1.f - f32[0]) * (j ? f32[1] : 1.f - f32[1]) * (k ? f32[2] : 1.f - f32[2]);
                    n[0] = n[0] + w32 * tn[3 * q]; n[1] = n[1] + w32 * tn[3 * q + 1]; n[2] = n[2] + w32 * tn[3 * q + 2];
                }
        S32 l = sqrt_(dot3(n, n));                     // .normalized(), mesh.py:110
        n[0] = n[0] / l; n[1] = n[1] / l; n[2] = n[2] / l;
    }
    qrot(rot32, n, n_out);
    return out;
}

template <class S64, class S32>
SMAC_HD bool collide_mixed_hybrid(const PrimTable<double>& T64, const PrimTable<float>& T32, const S64* st13, const S64* p_pos, S32* v_io,
                                  float p_mass, double dt, double life, S32* ext6) {
    // normalised inverse quaternion, once for both lookups
    S64 iq[4] = {st13[3], -st13[4], -st13[5], -st13[6]};
    {
        S64 nq = sqrt_(iq[0] * iq[0] + iq[1] * iq[1] + iq[2] * iq[2] + iq[3] * iq[3]);
        iq[0] = iq[0] / nq; iq[1] = iq[1] / nq; iq[2] = iq[2] / nq; iq[3] = iq[3] / nq;
    }
    S32 st32[13];
    for (int i = 0; i < 13; ++i) st32[i] = narrow_(st13[i]);
    S32 D[3], r[3], cv[3], in[3];
    S64 dist = prim_sdf_normal_hybrid(T64, T32, st13, iq, st32 + 3, p_pos, D);      // :141, :145
    if (!(val(dist) <= 5e-3)) return false;                                        // :142-143
    S32 p_v_in[3] = {v_io[0], v_io[1], v_io[2]};
    S32 p_v[3] = {v_io[0], v_io[1], v_io[2]};
    for (int i = 0; i < 3; ++i) r[i] = narrow_(p_pos[i] - st13[i]);                  // :146
    collider_v(st32, r, cv);                                                       // :147
    for (int i = 0; i < 3; ++i) in[i] = p_v[i] - cv[i];                            // :149
    S32 nc = dot3(in, D);                                                          // :150
    if (val(nc) < 0.f) {                                                           // :152
        S32 t[3] = {in[0] - nc * D[0], in[1] - nc * D[1], in[2] - nc * D[2]};      // :153
        S32 tt = dot3(t, t);
        S32 tn = sqrt_(tt + 1e-8f);                                                // length(), primitive_utils.py:5
        S32 scale = maxc(tn + nc * T32.friction, 0.f) / tn;                        // :155
        float flag = (std::sqrt(val(tt)) > 1e-30f) ? 1.f : 0.f;                    // :156
        for (int i = 0; i < 3; ++i) t[i] = (t[i] * scale) * flag + t[i] * (1.f - flag);   // :157
        if (val(dist) > 0.0) {                                                     // :161-163
            S32 infl = minc(exp_(-narrow_(dist) * T32.softness), 1.f);
            for (int i = 0; i < 3; ++i) p_v[i] = cv[i] + in[i] * (1.f - infl) + t[i] * infl;
        } else {
            for (int i = 0; i < 3; ++i) p_v[i] = cv[i] + t[i];                     // :159
        }
    }
    S64 xn[3] = {widen_(p_v[0]) * dt + p_pos[0], widen_(p_v[1]) * dt + p_pos[1], widen_(p_v[2]) * dt + p_pos[2]};   // :166
    S32 n2[3];
    S64 sdf2 = prim_sdf_normal_hybrid(T64, T32, st13, iq, st32 + 3, xn, n2);        // :167, :169
    if (val(sdf2) < 0.0) {                                                         // :168-170
        S32 k = narrow_(sdf2 * (life / dt));                                       // (the f64 distance times a constant: narrowing the PRODUCT costs 6e-8 of it)
        for (int i = 0; i < 3; ++i) p_v[i] = p_v[i] - k * n2[i];
    }
    S32 bf[3], bt[3];
    const float m_dt = p_mass * (float)(1.0 / dt);
    for (int i = 0; i < 3; ++i) bf[i] = (p_v_in[i] - p_v[i]) * m_dt;               // :173
    cross3(r, bf, bt);                                                             // :174
    for (int i = 0; i < 3; ++i) { ext6[i] = bf[i]; ext6[3 + i] = bt[i]; v_io[i] = p_v[i]; }
    return true;
}

// Adjoint of collide_mixed by 19 forward-mode passes (inputs: p_pos3, p_v3, state13).
// g_v[3]: adjoint of the output velocity, g_ext[6]: adjoint seeds of ext_f.
// Accumulates into g_pos[3], g_state[13]; OVERWRITES g_vin[3].
// Returns false (and touches nothing) when the particle is outside the band (identity map).
template <class R>
SMAC_HD bool collide_mixed_adjoint(const PrimTable<R>& T, const R* st13, const R* p_pos, const R* p_v,
                                   R p_mass, R dt, R life, const R* g_v, const R* g_ext,
                                   R* g_pos, R* g_vin, R* g_state) {
    {
        R d0 = prim_sdf(T, st13, p_pos);
        if (!(d0 <= R(5e-3))) return false;
    }
    R out[19];
    for (int dir = 0; dir < 19; ++dir) {
        Dual<R> pos[3], v[3], st[13], ext[6];
        for (int i = 0; i < 3; ++i) pos[i] = Dual<R>(p_pos[i], dir == i ? R(1) : R(0));
        for (int i = 0; i < 3; ++i) v[i] = Dual<R>(p_v[i], dir == 3 + i ? R(1) : R(0));
        for (int i = 0; i < 13; ++i) st[i] = Dual<R>(st13[i], dir == 6 + i ? R(1) : R(0));
        collide_mixed(T, st, pos, v, p_mass, dt, life, ext);
        R s = R(0);
        for (int i = 0; i < 3; ++i) s += g_v[i] * v[i].d;
        for (int i = 0; i < 6; ++i) s += g_ext[i] * ext[i].d;
        out[dir] = s;
    }
    for (int i = 0; i < 3; ++i) g_pos[i] += out[i];
    for (int i = 0; i < 3; ++i) g_vin[i] = out[3 + i];
    for (int i = 0; i < 13; ++i) g_state[i] += out[6 + i];
    return true;
}

// abs with the derivative of the taken branch
template <class R> SMAC_HD R abs_(R a) { return a < R(0) ? -a : a; }
template <class R> SMAC_HD Dual<R> abs_(Dual<R> a) { return a.v < R(0) ? -a : a; }

// collide_particle, primitive_base.py:105-137 (collision_type 1, called from p2g :203-206).
// Returns true when c = dist - 5e-3 < 0; imp3 = p_f * dt is the impulse added to the particle's momentum,
// ext6 this particle's wrench contribution.
template <class S, class R>
SMAC_HD bool collide_particle(const PrimTable<R>& T, const S* st13, const S* p_pos, const S* p_v, R dt, S* imp3, S* ext6) {
    S dist = prim_sdf(T, st13, p_pos);
    S c = dist - R(5e-3);                                                   // :108-109
    if (!(val(c) < R(0))) return false;
    S D[3], r[3], cv[3], in[3];
    prim_normal(T, st13, p_pos, D);
    for (int i = 0; i < 3; ++i) r[i] = p_pos[i] - st13[i];
    collider_v(st13, r, cv);
    for (int i = 0; i < 3; ++i) in[i] = p_v[i] - cv[i];
    S nc = dot3(in, D);
    S t[3] = {in[0] - nc * D[0], in[1] - nc * D[1], in[2] - nc * D[2]};     // :118
    S tn = sqrt_(dot3(t, t) + R(1e-8));                                     // :124
    S an = abs_(nc);
    S bf[3], bt[3];
    for (int i = 0; i < 3; ++i) {
        S f1 = -D[i] * c * R(50);                                           // :120-121
        S f2 = -(t[i] / tn) * an * T.friction;                              // :126
        imp3[i] = (f1 + f2) * dt;                                           // :128, 137
        bf[i] = -(f1 + f2);                                                 // :129
    }
    cross3(r, bf, bt);
    for (int i = 0; i < 3; ++i) { ext6[i] = bf[i]; ext6[3 + i] = bt[i]; }
    return true;
}

// collide (grid contact), primitive_base.py:72-103 (collision_type 0, called from grid_op :290-294).
// grid_pos is a node position (integer index * dx: carries no derivative), v_io the node velocity, grid_m its mass.
template <class S, class R>
SMAC_HD bool collide_grid(const PrimTable<R>& T, const S* st13, const R* grid_pos, S* v_io, S grid_m, R dt, S* ext6) {
    S pos[3] = {S(grid_pos[0]), S(grid_pos[1]), S(grid_pos[2])};
    S dist = prim_sdf(T, st13, pos);
    S infl = minc(exp_(-dist * T.softness), R(1));                         // :75
    if (!((T.softness > R(0) && val(infl) > R(0.1)) || val(dist) <= R(0))) return false;   // :76
    S v_in[3] = {v_io[0], v_io[1], v_io[2]};
    S D[3], r[3], cv[3], in[3];
    prim_normal(T, st13, pos, D);
    for (int i = 0; i < 3; ++i) r[i] = pos[i] - st13[i];
    collider_v(st13, r, cv);
    for (int i = 0; i < 3; ++i) in[i] = v_io[i] - cv[i];
    S nc = dot3(in, D);
    S ncm = minc(nc, R(0));                                                 // :86 ti.min(normal_component, 0)
    S t[3] = {in[0] - ncm * D[0], in[1] - ncm * D[1], in[2] - ncm * D[2]};
    S tt = dot3(t, t);
    S tn = sqrt_(tt + R(1e-8));
    S scale = maxc(tn + nc * T.friction, R(0)) / tn;                        // :89
    R flag = (val(nc) < R(0) && std::sqrt(val(tt)) > R(1e-30)) ? R(1) : R(0);   // :90
    S bf[3], bt[3];
    for (int i = 0; i < 3; ++i) {
        S t2 = (t[i] * scale) * flag + t[i] * (R(1) - flag);               // :91
        S vo = cv[i] + in[i] * (R(1) - infl) + t2 * infl;                   // :92
        bf[i] = grid_m * (v_in[i] - vo) * (R(1) / dt);                      // :95
        v_io[i] = vo;
    }
    cross3(r, bf, bt);
    for (int i = 0; i < 3; ++i) { ext6[i] = bf[i]; ext6[3 + i] = bt[i]; }
    return true;
}

// ------------------------------------------------------------------------------------------
// quadratic B-spline stencil, mpm_simulator.py:215-217
// ------------------------------------------------------------------------------------------
template <class R> struct Stencil {
    int base[3];
    R fx[3];
    R w[3][3];    // w[k][d]: weight of offset k along dimension d
    R dw[3][3];   // d w[k][d] / d fx[d]
};
template <class R> SMAC_HD void make_stencil(const R* x, R inv_dx, Stencil<R>& s) {
    for (int d = 0; d < 3; ++d) {
        R xs = x[d] * inv_dx;
        int b = (int)(xs - R(0.5));                 // .cast(int): truncation
        R fx = xs - R(b);
        s.base[d] = b; s.fx[d] = fx;
        s.w[0][d] = R(0.5) * (R(1.5) - fx) * (R(1.5) - fx);
        s.w[1][d] = R(0.75) - (fx - R(1)) * (fx - R(1));
        s.w[2][d] = R(0.5) * (fx - R(0.5)) * (fx - R(0.5));
        s.dw[0][d] = -(R(1.5) - fx);
        s.dw[1][d] = R(-2) * (fx - R(1));
        s.dw[2][d] = fx - R(0.5);
    }
}

// ------------------------------------------------------------------------------------------
// Particle positions.  R = double: plain doubles.  R = float: 32-bit FIXED POINT on [0, 1) carried in the 4-byte slots of
// the position rows - a resolution of 2^-32 = 2.3e-10 everywhere in the unit box, where a float has 6e-8 at x ~ 0.5.
// Same bytes, and it is what lets the f32 mode meet 1e-5: the stencil offset fx = x n - base is then exact to f32
// (a float x costs 4e-6 of a cell at n = 128), x += dt v accumulates increments of 1e-5 without losing them, and the
// forecast push-out (sdf / dt) n (primitive_base.py:170) divides a position error of 1e-10, not 3e-8, by dt.
// The reference keeps particles inside [3 dx, 1 - 3 dx] (boundary_condition :268-281); anything outside [0, 1) saturates.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Frame layout.  A frame holds NCOMP_ROWS components per particle.  SMAC_TILE_P = 0: SoA rows, S[c][p] (rows Npad apart).
// SMAC_TILE_P = T (a power of two): AoSoA tiles of T particles, S[p / T][c][p % T] - the 15 output streams of k_g2p then
// stay within a few DRAM pages (tools/microbench/frame_layout.hip: the 3-read / 15-write pattern streams at 2.1 TB/s from SoA
// rows and 3.4 TB/s from 1024-particle tiles; read-heavy patterns are unchanged or better).
//     element (c, p) of a frame = frame[rowbase(c, Npad) + poff(p)]
// ------------------------------------------------------------------------------------------
#ifndef SMAC_TILE_P
#define SMAC_TILE_P 0
#endif
constexpr int NCOMP_ROWS = 24;
SMAC_HD size_t rowbase(int c, int Npad) { return SMAC_TILE_P ? (size_t)c * (size_t)SMAC_TILE_P : (size_t)c * (size_t)Npad; }
SMAC_HD size_t poff(int p) {
    constexpr int T = SMAC_TILE_P ? SMAC_TILE_P : 1;
    return SMAC_TILE_P ? (size_t)(p / T) * (size_t)(NCOMP_ROWS * T) + (size_t)(p % T) : (size_t)p;
}
SMAC_HD size_t rowoff(int c, int p, int Npad) { return rowbase(c, Npad) + poff(p); }

template <class R> struct pos_of { typedef R type; };
template <> struct pos_of<float> { typedef uint32_t type; };
SMAC_HD double pos_get(double x) { return x; }
SMAC_HD double pos_get(uint32_t x) { return (double)x * (1.0 / 4294967296.0); }
SMAC_HD void pos_set(double v, double& o) { o = v; }
SMAC_HD void pos_set(double v, uint32_t& o) {
    const double s = v * 4294967296.0 + 0.5;
    o = !(s > 0.0) ? 0u : (s >= 4294967295.0 ? 4294967295u : (uint32_t)s);          // NaN -> 0
}
// base = int(x n - 0.5) (truncation, :215) and fx = x n - base of a stored position
SMAC_HD void pos_base_fx(double x, int n, int& b, double& fx) {
    const double xs = x * (double)n;
    b = (int)(xs - 0.5);
    fx = xs - (double)b;
}
#ifndef SMAC_POS_INT
#define SMAC_POS_INT 1
#endif
SMAC_HD void pos_base_fx(uint32_t x, int n, int& b, float& fx) {
#if !SMAC_POS_INT
    const double xs = pos_get(x) * (double)n;
    b = (int)(xs - 0.5);
    fx = (float)(xs - (double)b);
    return;
#endif
    // fixed point: x n is a 64-bit integer with 32 fraction bits - exact, and no f64 instruction
    const uint64_t prod = (uint64_t)x * (uint64_t)(uint32_t)n;
    if (prod < 0x80000000ull) { b = 0; fx = (float)(uint32_t)prod * 2.3283064365386963e-10f; return; }    // x n < 0.5
    const uint64_t q = prod - 0x80000000ull;
    b = (int)(q >> 32);
    fx = (float)(uint32_t)q * 2.3283064365386963e-10f + 0.5f;
}
SMAC_HD int pos_base(double x, int n) { return (int)(x * (double)n - 0.5); }
SMAC_HD int pos_base(uint32_t x, int n) {
    const uint64_t prod = (uint64_t)x * (uint64_t)(uint32_t)n;
    return prod < 0x80000000ull ? 0 : (int)((prod - 0x80000000ull) >> 32);
}
// x + dt v of g2p (:318)
SMAC_HD double pos_advance(double x, double dt, double v) { return x + dt * v; }
SMAC_HD uint32_t pos_advance(uint32_t x, double dt, float v) {
    // the increment in fixed-point units (|dt v| < 2^-8 of the box for any admissible step); saturating add
    const float d = (float)(dt * 4294967296.0) * v;
    const long long y = (long long)x + (long long)(d < 0.f ? d - 0.5f : d + 0.5f);
    return y <= 0 ? 0u : (y >= 4294967295ll ? 4294967295u : (uint32_t)y);
}
// the stencil of mpm_simulator.py:215-217 from a stored position (n = n_grid = inv_dx)
template <class R, class P> SMAC_HD void make_stencil_pos(const P* x, int n, Stencil<R>& s) {
    for (int d = 0; d < 3; ++d) {
        int b;
        R fx;
        pos_base_fx(x[d], n, b, fx);
        s.base[d] = b; s.fx[d] = fx;
        s.w[0][d] = R(0.5) * (R(1.5) - fx) * (R(1.5) - fx);
        s.w[1][d] = R(0.75) - (fx - R(1)) * (fx - R(1));
        s.w[2][d] = R(0.5) * (fx - R(0.5)) * (fx - R(0.5));
        s.dw[0][d] = -(R(1.5) - fx);
        s.dw[1][d] = R(-2) * (fx - R(1));
        s.dw[2][d] = fx - R(0.5);
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 helpers (row-major R[9])
// ------------------------------------------------------------------------------------------
template <class R> SMAC_HD void mm(const R* A, const R* B, R* C) {          // C = A B
    SMAC_PRECISE_FP
    R t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    for (int i = 0; i < 9; ++i) C[i] = t[i];
}
template <class R> SMAC_HD void mtm(const R* A, const R* B, R* C) {         // C = A^T B
    SMAC_PRECISE_FP
    R t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
    for (int i = 0; i < 9; ++i) C[i] = t[i];
}
template <class R> SMAC_HD void mmt(const R* A, const R* B, R* C) {         // C = A B^T
    SMAC_PRECISE_FP
    R t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
    for (int i = 0; i < 9; ++i) C[i] = t[i];
}

// det(I + E) - 1, evaluated without the cancellation of det(F) - 1
template <class R> SMAC_HD R det_minus_one(const R* E) {
    SMAC_PRECISE_FP
    R tr = E[0] + E[4] + E[8];
    R m2 = (E[0] * E[4] - E[1] * E[3]) + (E[0] * E[8] - E[2] * E[6]) + (E[4] * E[8] - E[5] * E[7]);
    R d3 = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
    return tr + m2 + d3;
}
// cofactor matrix of F = I + E:  cof(F) = dF det
template <class R> SMAC_HD void cofactor(const R* E, R* K) {
    SMAC_PRECISE_FP
    R F[9];
    for (int i = 0; i < 9; ++i) F[i] = E[i];
    F[0] += R(1); F[4] += R(1); F[8] += R(1);
    K[0] = F[4] * F[8] - F[5] * F[7]; K[1] = F[5] * F[6] - F[3] * F[8]; K[2] = F[3] * F[7] - F[4] * F[6];
    K[3] = F[2] * F[7] - F[1] * F[8]; K[4] = F[0] * F[8] - F[2] * F[6]; K[5] = F[1] * F[6] - F[0] * F[7];
    K[6] = F[1] * F[5] - F[2] * F[4]; K[7] = F[2] * F[3] - F[0] * F[5]; K[8] = F[0] * F[4] - F[1] * F[3];
}

// Jacobi stops when the off-diagonal mass is at the rounding floor of the rotations themselves
// (a few ulp of |H|); asking for less only burns sweeps.  The f32 rotations use the hardware
// reciprocal / rsqrt (1 ulp): a rotation that is 1e-7 off is corrected by the next sweep.
template <class R> struct eps_of;
template <> struct eps_of<float> { static constexpr float v = 4.0e-7f; static constexpr int sweeps = 6; };
template <> struct eps_of<double> { static constexpr double v = 1.0e-15; static constexpr int sweeps = 10; };

SMAC_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
SMAC_HD double fast_rcp(double x) { return 1.0 / x; }
SMAC_HD float fast_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / std::sqrt(x);
#endif
}
SMAC_HD double fast_rsqrt(double x) { return 1.0 / std::sqrt(x); }
SMAC_HD float fast_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return std::sqrt(x);
#endif
}
SMAC_HD double fast_sqrt(double x) { return std::sqrt(x); }

// One Jacobi rotation on the symmetric matrix {a00,a01,a02,a11,a12,a22} in the (p,q) plane,
// accumulated into V (columns = eigenvectors).  Written out per pair to keep everything in registers.
// tan 2θ = b / d with d = aqq - app, b = 2 apq.  With r = |(d, b)| and the small-angle root:
//     cos²θ = (1 + |d| / r) / 2,     sinθ cosθ = sgn(d) b / (2 r),     t = tanθ = sinθ / cosθ
// which takes two reciprocal square roots where the textbook form (tau = d / b, t = sgn / (|tau| + sqrt(1 + tau²)), c = rsqrt(1 + t²))
// takes four quarter-rate operations; no quantity is a difference of nearly equal numbers, so s and t keep full relative precision
// for the tiny angles of the last sweep.  c² + s² = 1 to the rounding of the second rsqrt, as before.
template <class R> struct tiny_of;
template <> struct tiny_of<float> { static constexpr float v = 1.0e-30f, huge = 1.0e37f; };
template <> struct tiny_of<double> { static constexpr double v = 1.0e-280, huge = 1.0e300; };
template <class R> SMAC_HD void jacobi_cs(R app, R aqq, R apq, R& c, R& s, R& t) {
    SMAC_PRECISE_FP
    const R d = aqq - app, b = R(2) * apq;
    const R rr = d * d + b * b;
    // apq = 0: nothing to rotate.  rr below the underflow guard (strains under 1e-15): the rotation would change nothing representable;
    // rr overflowed (an exploded state): left alone, as the textbook form did for a huge tau.
    if (apq == R(0) || !(rr > tiny_of<R>::v && rr < tiny_of<R>::huge)) { c = R(1); s = R(0); t = R(0); return; }
    const R rinv = fast_rsqrt(rr);
    const R ad = d < R(0) ? -d : d;
    const R c2 = R(0.5) + R(0.5) * (ad * rinv);
    const R cinv = fast_rsqrt(c2);
    c = c2 * cinv;
    s = (R(0.5) * b * rinv) * cinv;
    if (d < R(0)) s = -s;
    t = s * cinv;
}

// SVD of F = I + E for the constitutive model: F = U diag(1+e) V^T.
// Works on H = F^T F - I = E + E^T + E^T E so that e_i = sigma_i - 1 keeps full relative
// precision for the small strains of stiff materials (fp32 storage).
// For det F < 0 the sign moves into the smallest singular value (ti.svd contract: U, V rotations
// up to a common sign, which every consumer ignores).
template <class R> SMAC_HD void svd_I_plus_E(const R* E, R* U, R* e, R* V) {
    SMAC_PRECISE_FP
    R a00 = R(2) * E[0] + E[0] * E[0] + E[3] * E[3] + E[6] * E[6];
    R a11 = R(2) * E[4] + E[1] * E[1] + E[4] * E[4] + E[7] * E[7];
    R a22 = R(2) * E[8] + E[2] * E[2] + E[5] * E[5] + E[8] * E[8];
    R a01 = E[1] + E[3] + E[0] * E[1] + E[3] * E[4] + E[6] * E[7];
    R a02 = E[2] + E[6] + E[0] * E[2] + E[3] * E[5] + E[6] * E[8];
    R a12 = E[5] + E[7] + E[1] * E[2] + E[4] * E[5] + E[7] * E[8];
    for (int i = 0; i < 9; ++i) V[i] = R(0);
    V[0] = V[4] = V[8] = R(1);
    for (int sweep = 0; sweep < eps_of<R>::sweeps; ++sweep) {
        R off = std::fabs(a01) + std::fabs(a02) + std::fabs(a12);
        R dia = std::fabs(a00) + std::fabs(a11) + std::fabs(a22);
        if (off <= eps_of<R>::v * dia || off == R(0)) break;
        R c, s, t;
        // (0,1)
        jacobi_cs(a00, a11, a01, c, s, t);
        { R n00 = a00 - t * a01, n11 = a11 + t * a01;
          R n02 = c * a02 - s * a12, n12 = s * a02 + c * a12;
          a00 = n00; a11 = n11; a01 = R(0); a02 = n02; a12 = n12;
          for (int k = 0; k < 3; ++k) { R p = V[3 * k], q = V[3 * k + 1]; V[3 * k] = c * p - s * q; V[3 * k + 1] = s * p + c * q; } }
        // (0,2)
        jacobi_cs(a00, a22, a02, c, s, t);
        { R n00 = a00 - t * a02, n22 = a22 + t * a02;
          R n01 = c * a01 - s * a12, n12 = s * a01 + c * a12;
          a00 = n00; a22 = n22; a02 = R(0); a01 = n01; a12 = n12;
          for (int k = 0; k < 3; ++k) { R p = V[3 * k], q = V[3 * k + 2]; V[3 * k] = c * p - s * q; V[3 * k + 2] = s * p + c * q; } }
        // (1,2)
        jacobi_cs(a11, a22, a12, c, s, t);
        { R n11 = a11 - t * a12, n22 = a22 + t * a12;
          R n01 = c * a01 - s * a02, n02 = s * a01 + c * a02;
          a11 = n11; a22 = n22; a12 = R(0); a01 = n01; a02 = n02;
          for (int k = 0; k < 3; ++k) { R p = V[3 * k + 1], q = V[3 * k + 2]; V[3 * k + 1] = c * p - s * q; V[3 * k + 2] = s * p + c * q; } }
    }
    R h[3] = {a00, a11, a22};
    // B = F V = V + E V ; sigma_i = sqrt(1 + h_i) ; U_i = B_i / sigma_i
    R B[9];
    mm(E, V, B);
    for (int i = 0; i < 9; ++i) B[i] += V[i];
    for (int i = 0; i < 3; ++i) {
        R hi = h[i] > R(-1) ? h[i] : R(-1);
        R s2 = R(1) + hi;
        R inv = s2 > R(1e-38) ? fast_rsqrt(s2) : R(0);              // 1 / sigma_i (one rsqrt gives sigma_i and its reciprocal)
        R sg = s2 * inv;
        e[i] = hi * fast_rcp(R(1) + sg);
        U[i] = B[i] * inv; U[3 + i] = B[3 + i] * inv; U[6 + i] = B[6 + i] * inv;
    }
    // inverted element: flip the smallest singular value and its U column
    R detm1 = det_minus_one(E);
    if (detm1 < R(-1)) {
        int k = 0;
        if (e[1] < e[k]) k = 1;
        if (e[2] < e[k]) k = 2;
        for (int j = 0; j < 3; ++j)
            if (j == k) { e[j] = -(R(2) + e[j]); U[j] = -U[j]; U[3 + j] = -U[3 + j]; U[6 + j] = -U[6 + j]; }
    }
}

// ------------------------------------------------------------------------------------------
// constitutive update (mpm_simulator.py:219-250) on E = F - I.
//   in : Et = F_tmp - I
//   out: En = new_F - I, stress (un-scaled, reference line 235-236 / 244-245)
//   kept for the adjoint: U, e, V, ep (clipped e), Jm1 = J - 1
// ------------------------------------------------------------------------------------------
// plast: how the plastic material (ptype 0) returns to the yield surface.  PLAST_CLIP: the singular values are clipped to
// [1 - 2e-3, 1 + 3e-3] (softmac mpm_simulator.py:226-229).  PLAST_VON_MISES: soft_cloth's compute_von_mises
// (soft_cloth/engine/mpm_simulator.py:172-188) with yield_c = yield_stress / (2 mu).
enum { PLAST_CLIP = 0, PLAST_VON_MISES = 1 };
template <class R> struct Material {
    int ptype, model;
    R mu, lam;
    int plast;
    R yield_c;
};
// von-Mises return mapping on e = sigma - 1 (log strains eps = log(max(sigma, 0.05)), deviator eh, |eh| with the reference's 1e-8):
// returns whether the particle yields; ep = sigma' - 1
template <class R> SMAC_HD bool von_mises(const R* e, R yield_c, R* ep, R* eh, R& nrm) {
    R eps[3];
    for (int i = 0; i < 3; ++i) eps[i] = std::log1p(max_(e[i], R(0.05 - 1.0)));           // :175, 177-179
    const R m = (eps[0] + eps[1] + eps[2]) / R(3);
    for (int i = 0; i < 3; ++i) eh[i] = eps[i] - m;                                         // :180
    nrm = std::sqrt(eh[0] * eh[0] + eh[1] * eh[1] + eh[2] * eh[2] + R(1e-8));               // :181, 200-202
    const R dg = nrm - yield_c;                                                             // :182
    for (int i = 0; i < 3; ++i) ep[i] = e[i];
    if (!(dg > R(0))) return false;
    for (int i = 0; i < 3; ++i) ep[i] = std::expm1(eps[i] - (dg / nrm) * eh[i]);            // :185-186
    return true;
}
template <class R> struct ConstState {
    R U[9], V[9], e[3], ep[3], Jm1;
    bool has_svd;
};

template <class R>
SMAC_HD void constitutive_fwd(const Material<R>& M, const R* Et, R* En, R* stress, ConstState<R>& cs) {
    SMAC_PRECISE_FP
    cs.Jm1 = det_minus_one(Et);                                        // :222
    const R J = R(1) + cs.Jm1;
    cs.has_svd = false;
    for (int i = 0; i < 9; ++i) { En[i] = Et[i]; stress[i] = R(0); }
    if (M.model == MODEL_COROTATED) {
        if (M.ptype == MAT_LIQUID) {                                   // :233  new_F = J^(1/3) I
            R c = std::cbrt(J);
            R cm1 = cs.Jm1 / (c * c + c + R(1));                       // c - 1 without cancellation
            for (int i = 0; i < 9; ++i) En[i] = R(0);
            En[0] = En[4] = En[8] = cm1;
            if (M.mu != R(0)) {                                        // 2 mu (cI - R) c ; needs R = U V^T
                svd_I_plus_E(Et, cs.U, cs.e, cs.V);
                cs.has_svd = true;
                R Rm[9];
                mmt(cs.U, cs.V, Rm);
                for (int i = 0; i < 9; ++i) stress[i] = -R(2) * M.mu * c * Rm[i];
                stress[0] += R(2) * M.mu * c * c; stress[4] += R(2) * M.mu * c * c; stress[8] += R(2) * M.mu * c * c;
            }
        } else {
            svd_I_plus_E(Et, cs.U, cs.e, cs.V);                        // :130-133
            cs.has_svd = true;
            for (int i = 0; i < 3; ++i) {
                cs.ep[i] = cs.e[i];
                if (M.ptype == MAT_PLASTIC && M.plast == PLAST_CLIP)   // :226-229
                    cs.ep[i] = min_(max_(cs.e[i], R(-2e-3)), R(3e-3));
            }
            if (M.ptype == MAT_PLASTIC && M.plast == PLAST_VON_MISES) {
                R eh[3], nrm;
                von_mises(cs.e, M.yield_c, cs.ep, eh, nrm);
            }
            // (new_F - R) new_F^T = U diag(e'(1+e')) U^T              (:234-235)
            R d[3] = {cs.ep[0] * (R(1) + cs.ep[0]), cs.ep[1] * (R(1) + cs.ep[1]), cs.ep[2] * (R(1) + cs.ep[2])};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    stress[3 * i + j] = R(2) * M.mu * (cs.U[3 * i] * d[0] * cs.U[3 * j] + cs.U[3 * i + 1] * d[1] * cs.U[3 * j + 1] +
                                                       cs.U[3 * i + 2] * d[2] * cs.U[3 * j + 2]);
            if (M.ptype == MAT_PLASTIC) {                              // new_F = F_tmp + U diag(e'-e) V^T
                R dd[3] = {cs.ep[0] - cs.e[0], cs.ep[1] - cs.e[1], cs.ep[2] - cs.e[2]};
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        En[3 * i + j] += cs.U[3 * i] * dd[0] * cs.V[3 * j] + cs.U[3 * i + 1] * dd[1] * cs.V[3 * j + 1] +
                                         cs.U[3 * i + 2] * dd[2] * cs.V[3 * j + 2];
            }
        }
        R p = M.lam * J * cs.Jm1;                                      // :236
        stress[0] += p; stress[4] += p; stress[8] += p;
    } else {                                                           // :237-245 neo-Hookean
        R Fn[9];
        if (M.ptype == MAT_LIQUID) {                                   // :242-243
            R sq = std::sqrt(J);
            for (int i = 0; i < 9; ++i) En[i] = R(0);
            En[0] = En[4] = cs.Jm1 / (sq + R(1));
        }
        for (int i = 0; i < 9; ++i) Fn[i] = En[i];
        // mu (F F^T - I) + lam log(J) I  ==  mu F F^T + (lam log J - mu) I
        R FFt[9];
        mmt(Fn, Fn, FFt);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) stress[3 * i + j] = M.mu * (Fn[3 * i + j] + Fn[3 * j + i] + FFt[3 * i + j]);
        R p = M.lam * std::log1p(cs.Jm1);
        stress[0] += p; stress[4] += p; stress[8] += p;
    }
}

// reference clamp(), mpm_simulator.py:184-192
template <class R> SMAC_HD R clamp_ref(R a) { return a >= R(0) ? max_(a, R(1e-6)) : min_(a, R(-1e-6)); }

// Adjoint of constitutive_fwd.  G = adjoint of the un-scaled stress, gFn = adjoint of new_F
// (= F.grad[f+1]).  Output gEt = adjoint of F_tmp, following p2g.grad + svd_grad
// (mpm_simulator.py:371-373) including backward_svd's clamp.
template <class R>
SMAC_HD void constitutive_bwd(const Material<R>& M, const R* Et, const ConstState<R>& cs,
                              const R* G, const R* gFn, R* gEt) {
    SMAC_PRECISE_FP
    const R J = R(1) + cs.Jm1;
    R gJ = R(0);
    for (int i = 0; i < 9; ++i) gEt[i] = R(0);
    if (M.model == MODEL_COROTATED) {
        gJ = M.lam * (R(2) * J - R(1)) * (G[0] + G[4] + G[8]);         // d/dJ lam J (J-1) tr
        if (M.ptype == MAT_LIQUID) {
            R c = std::cbrt(J);
            // new_F = c I  -> gc = tr(gFn) ; dc/dJ = c / (3 J)
            R gc = gFn[0] + gFn[4] + gFn[8];
            if (M.mu != R(0)) {
                // stress_mu = 2 mu (c^2 I - c R): adjoint wrt c and wrt R, then R through the SVD adjoint
                R Rm[9];
                mmt(cs.U, cs.V, Rm);
                R trG = G[0] + G[4] + G[8], GR = R(0);
                for (int i = 0; i < 9; ++i) GR += G[i] * Rm[i];
                gc += R(2) * M.mu * (R(2) * c * trG - GR);
                // B = adjoint of R = -2 mu c G ; MB = U^T B V
                R MB[9], t[9];
                mtm(cs.U, G, t); mm(t, cs.V, MB);
                for (int i = 0; i < 9; ++i) MB[i] *= -R(2) * M.mu * c;
                R T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        if (i != j) {
                            R ds = cs.e[j] - cs.e[i];
                            R K = R(1) / clamp_ref(ds * (R(2) + cs.e[i] + cs.e[j]));
                            T[3 * i + j] = K * (MB[3 * i + j] - MB[3 * j + i]) * ds;
                        }
                mm(cs.U, T, t); mmt(t, cs.V, gEt);
            }
            gJ += gc * c / (R(3) * J);
        } else {
            // Ghat = U^T G U ; N = U^T gFn V
            R Gh[9], N[9], t[9];
            mtm(cs.U, G, t); mm(t, cs.U, Gh);
            mtm(cs.U, gFn, t); mm(t, cs.V, N);
            const R* e = cs.e; const R* ep = cs.ep;
            R sp[3] = {R(1) + ep[0], R(1) + ep[1], R(1) + ep[2]};
            // stress part of MA (adjoint of new_F in the U,V basis) and MB (adjoint of R)
            R MAs[9], MB[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    MAs[3 * i + j] = R(2) * M.mu * (Gh[3 * i + j] * sp[j] + Gh[3 * j + i] * ep[j]);
                    MB[3 * i + j] = -R(2) * M.mu * Gh[3 * i + j] * sp[j];
                }
            R T[9];
            const bool plastic = (M.ptype == MAT_PLASTIC);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    R ma_ij = plastic ? MAs[3 * i + j] + N[3 * i + j] : R(0);
                    R ma_ji = plastic ? MAs[3 * j + i] + N[3 * j + i] : R(0);
                    if (i == j) {
                        bool inside = (e[i] > R(-2e-3)) && (e[i] < R(3e-3));     // clip passes the gradient
                        T[3 * i + j] = (plastic && (inside || M.plast != PLAST_CLIP)) ? ma_ij : R(0);   // (von Mises: adjoint of sigma'_i, mapped below)
                    } else {
                        // reference: K (ma_ij (a_j - a_i) + ma_ji b + (MB_ij - MB_ji) ds) with a = s s' - 1, b = s_i s'_j - s'_i s_j.
                        // Written on the differences ds = e_j - e_i, ds' = e'_j - e'_i:
                        //     a_j - a_i = ds (1 + e'_i) + ds' (1 + e_j),     b = ds' (1 + e_i) - ds (1 + e'_i),
                        // the bracket is K ds [...] + K ds' [...], and K ds = 1 / (2 + e_i + e_j) outside the clamp: the
                        // rounding of nearly equal strains cancels instead of being multiplied by K (up to 1e6)
                        R ds = e[j] - e[i], dsp = ep[j] - ep[i];
                        R K = R(1) / clamp_ref(ds * (R(2) + e[i] + e[j]));       // 1/clamp(s_j^2 - s_i^2)
                        T[3 * i + j] = (K * ds) * ((ma_ij - ma_ji) * (R(1) + ep[i]) + (MB[3 * i + j] - MB[3 * j + i])) +
                                       (K * dsp) * (ma_ij * (R(1) + e[j]) + ma_ji * (R(1) + e[i]));
                    }
                    if (!plastic) T[3 * i + j] += MAs[3 * i + j];      // elastic: new_F = F_tmp directly
                }
            if (plastic && M.plast == PLAST_VON_MISES) {
                // the diagonal of T holds the adjoints of sigma'_k; sigma' = exp(eps'), eps' = mean(eps) + (c / |eh|) eh, eps = log(max(sigma, 0.05)):
                //   d eps'_k / d eps_i = 1/3 + (c / |eh|) (delta_ki - 1/3) - c eh_k eh_i / |eh|^3        (sum eh = 0)
                R ep2[3], eh[3], nrm;
                if (von_mises(e, M.yield_c, ep2, eh, nrm)) {             // (not yielding: new_F = F_tmp, the adjoint passes unchanged)
                    const R c = M.yield_c;
                    R gep[3] = {T[0] * (R(1) + ep[0]), T[4] * (R(1) + ep[1]), T[8] * (R(1) + ep[2])};
                    const R gm = (gep[0] + gep[1] + gep[2]) / R(3), dot = gep[0] * eh[0] + gep[1] * eh[1] + gep[2] * eh[2];
                    for (int i = 0; i < 3; ++i) {
                        const R ge = gm + (c / nrm) * (gep[i] - gm) - c * dot * eh[i] / (nrm * nrm * nrm);
                        T[4 * i] = (R(1) + e[i] > R(0.05)) ? ge / (R(1) + e[i]) : R(0);
                    }
                }
            }
            mm(cs.U, T, t); mmt(t, cs.V, gEt);
            if (!plastic) for (int i = 0; i < 9; ++i) gEt[i] += gFn[i];
        }
    } else {
        // stress = mu (Fn + Fn^T + Fn Fn^T - ... ) written on En ; Fn = I + En
        R Fn[9], En[9];
        for (int i = 0; i < 9; ++i) En[i] = Et[i];
        R gEn[9];
        if (M.ptype == MAT_LIQUID) {
            R sq = std::sqrt(J);
            for (int i = 0; i < 9; ++i) En[i] = R(0);
            En[0] = En[4] = cs.Jm1 / (sq + R(1));
        }
        for (int i = 0; i < 9; ++i) Fn[i] = En[i];
        Fn[0] += R(1); Fn[4] += R(1); Fn[8] += R(1);
        // d/dFn <G, mu Fn Fn^T> = mu (G + G^T) Fn
        R Gs[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Gs[3 * i + j] = M.mu * (G[3 * i + j] + G[3 * j + i]);
        mm(Gs, Fn, gEn);
        for (int i = 0; i < 9; ++i) gEn[i] += gFn[i];
        gJ = M.lam * (G[0] + G[4] + G[8]) / J;
        if (M.ptype == MAT_LIQUID) {
            gJ += (gEn[0] + gEn[4]) / (R(2) * std::sqrt(J));
        } else {
            for (int i = 0; i < 9; ++i) gEt[i] = gEn[i];
        }
    }
    R K[9];
    cofactor(Et, K);
    for (int i = 0; i < 9; ++i) gEt[i] += gJ * K[i];
}

}  // namespace smac
