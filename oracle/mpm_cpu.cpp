// CPU oracle #2 / CPU baseline: plain C++17 + OpenMP, float64.  TEST INFRASTRUCTURE ONLY - only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file; the product
// (softmac_amd/) never does.
//
// It restates the reference's substep with the reference's OWN decomposition - AOS particle arrays, a
// dense n^3 grid swept in full, one pass per Taichi kernel, atomics for the scatters:
//   clear_grid :93-114, compute_F_tmp :125-128, svd :130-133, p2g :198-262, grid_op_mixed1..4 :396-443,
//   g2p :299-318 and, for substep_grad :339-378, the same forward passes followed by the adjoint of each
//   kernel in reverse order, with svd_grad/backward_svd :135-157 taken literally (U, sig, V adjoints, clamp).
// (/root/reference/softmac/engine/mpm_simulator.py; contact: primitive/primitive_base.py:53-181,
//  primitive/mesh.py:45-113, primitive/primitive_utils.py:3-46.)
// PARITY UNPINNED at the same third-party boundaries as oracle/softmac_oracle.py (ti.svd, Taichi AD,
// literal typing); it is validated against that file by tests/test_cpu_port.py.
// The contact adjoint uses forward-mode duals (19 inputs), the rest is hand-written reverse mode.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

struct Params {
    int N, n, substeps, ptype, model, P, sticky, collision_type;
    double dt, mu, lam, p_vol, p_mass, g[3];
};
struct Prim {
    const double* sdf; const double* normal;
    int res[3];
    double lower[3], upper[3], inv_dx, friction, softness;
    int contact;
};

// ---------------------------------------------------------------- dual numbers for the contact adjoint
struct Dual { double v, d; Dual(double a = 0, double b = 0) : v(a), d(b) {} };
inline Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
inline Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
inline Dual operator-(Dual a) { return {-a.v, -a.d}; }
inline Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
inline Dual operator/(Dual a, Dual b) { double q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
inline double val(double a) { return a; }
inline double val(Dual a) { return a.v; }
inline double sqrt_(double a) { return std::sqrt(a); }
inline Dual sqrt_(Dual a) { double s = std::sqrt(a.v); return {s, a.d / (2 * s)}; }
inline double exp_(double a) { return std::exp(a); }
inline Dual exp_(Dual a) { double e = std::exp(a.v); return {e, e * a.d}; }

template <class S> void cross(const S* a, const S* b, S* o) {
    S x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
template <class S> void qrot(const S* q, const S* v, S* o) {                      // primitive_utils.py:7-13
    S uv[3], uuv[3];
    cross(q + 1, v, uv); cross(q + 1, uv, uuv);
    for (int i = 0; i < 3; ++i) o[i] = v[i] + S(2.0) * (q[0] * uv[i] + uuv[i]);
}
template <class S> void inv_trans(const S* pos, const S* p, const S* q, S* o) {   // :42-46
    S iq[4] = {q[0], -q[1], -q[2], -q[3]};
    S n = sqrt_(iq[0] * iq[0] + iq[1] * iq[1] + iq[2] * iq[2] + iq[3] * iq[3]);
    for (int i = 0; i < 4; ++i) iq[i] = iq[i] / n;
    S d[3] = {pos[0] - p[0], pos[1] - p[1], pos[2] - p[2]};
    qrot(iq, d, o);
}
template <class S> bool locate(const Prim& T, const S* local, int* b, S* fx) {   // mesh.py:50-58
    for (int i = 0; i < 3; ++i)
        if (val(local[i]) < T.lower[i] || val(local[i]) >= T.upper[i]) return false;
    for (int i = 0; i < 3; ++i) {
        S p = (local[i] - S(T.lower[i])) * S(T.inv_dx);
        b[i] = std::min((int)val(p), T.res[i] - 2);
        fx[i] = p - S((double)b[i]);
    }
    return true;
}
template <class S> S sdf_at(const Prim& T, const S* st, const S* pos) {           // primitive_base.py:53-56, mesh.py:45-68
    S local[3]; inv_trans(pos, st, st + 3, local);
    int b[3]; S fx[3];
    if (!locate(T, local, b, fx)) return S(1e10);
    S out(0.0);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
        S w = (i ? fx[0] : S(1.0) - fx[0]) * (j ? fx[1] : S(1.0) - fx[1]) * (k ? fx[2] : S(1.0) - fx[2]);
        out = out + w * S(T.sdf[((b[0] + i) * T.res[1] + (b[1] + j)) * T.res[2] + (b[2] + k)]);
    }
    return out;
}
template <class S> void normal_at(const Prim& T, const S* st, const S* pos, S* o) {   // :58-61, mesh.py:90-113
    S local[3]; inv_trans(pos, st, st + 3, local);
    int b[3]; S fx[3];
    S n[3] = {S(0.0), S(1.0), S(0.0)};
    if (locate(T, local, b, fx)) {
        n[1] = S(0.0);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
            S w = (i ? fx[0] : S(1.0) - fx[0]) * (j ? fx[1] : S(1.0) - fx[1]) * (k ? fx[2] : S(1.0) - fx[2]);
            const double* t = T.normal + 3 * (((b[0] + i) * T.res[1] + (b[1] + j)) * T.res[2] + (b[2] + k));
            for (int c = 0; c < 3; ++c) n[c] = n[c] + w * S(t[c]);
        }
        S l = sqrt_(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        for (int c = 0; c < 3; ++c) n[c] = n[c] / l;
    }
    qrot(st + 3, n, o);
}
template <class S> void collider_v(const S* st, const S* r, S* o) {               // primitive_base.py:63-70
    const S* q0 = st + 3;
    S n = sqrt_(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
    S q[4] = {q0[0] / n, q0[1] / n, q0[2] / n, q0[3] / n}, iq[4] = {q[0], -q[1], -q[2], -q[3]};
    S rl[3], wl[3], vl[3];
    qrot(iq, r, rl); cross(st + 10, rl, wl);
    for (int i = 0; i < 3; ++i) vl[i] = st[7 + i] + wl[i];
    qrot(q, vl, o);
}
// collide_mixed :139-181
template <class S> bool collide_mixed(const Prim& T, const S* st, const S* x, S* v, double p_mass, double dt, double life, S* ext) {
    S dist = sdf_at(T, st, x);
    if (!(val(dist) <= 5e-3)) return false;
    S vin[3] = {v[0], v[1], v[2]}, pv[3] = {v[0], v[1], v[2]}, D[3], r[3], cv[3], in[3];
    normal_at(T, st, x, D);
    for (int i = 0; i < 3; ++i) r[i] = x[i] - st[i];
    collider_v(st, r, cv);
    for (int i = 0; i < 3; ++i) in[i] = pv[i] - cv[i];
    S nc = in[0] * D[0] + in[1] * D[1] + in[2] * D[2];
    if (val(nc) < 0) {
        S t[3] = {in[0] - nc * D[0], in[1] - nc * D[1], in[2] - nc * D[2]};
        S tt = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
        S tn = sqrt_(tt + S(1e-8));
        S a = tn + nc * S(T.friction);
        S scale = (val(a) >= 0 ? a : S(0.0)) / tn;
        double flag = std::sqrt(val(tt)) > 1e-30 ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) t[i] = (t[i] * scale) * S(flag) + t[i] * S(1.0 - flag);
        if (val(dist) > 0) {
            S e = exp_(-dist * S(T.softness));
            S infl = val(e) <= 1.0 ? e : S(1.0);
            for (int i = 0; i < 3; ++i) pv[i] = cv[i] + in[i] * (S(1.0) - infl) + t[i] * infl;
        } else {
            for (int i = 0; i < 3; ++i) pv[i] = cv[i] + t[i];
        }
    }
    S xn[3] = {pv[0] * S(dt) + x[0], pv[1] * S(dt) + x[1], pv[2] * S(dt) + x[2]};
    S s2 = sdf_at(T, st, xn);
    if (val(s2) < 0) {
        S n2[3]; normal_at(T, st, xn, n2);
        S k = (s2 / S(dt)) * S(life);
        for (int i = 0; i < 3; ++i) pv[i] = pv[i] - k * n2[i];
    }
    S bf[3], bt[3];
    for (int i = 0; i < 3; ++i) bf[i] = (vin[i] - pv[i]) * S(p_mass * (1.0 / dt));
    cross(r, bf, bt);
    for (int i = 0; i < 3; ++i) { ext[i] = bf[i]; ext[3 + i] = bt[i]; v[i] = pv[i]; }
    return true;
}

// ---------------------------------------------------------------- 3x3 helpers (row-major)
inline void mm(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    std::memcpy(C, t, sizeof t);
}
inline void tr(const double* A, double* T) { double t[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]}; std::memcpy(T, t, sizeof t); }
inline double det3(const double* F) {
    return F[0] * (F[4] * F[8] - F[5] * F[7]) - F[1] * (F[3] * F[8] - F[5] * F[6]) + F[2] * (F[3] * F[7] - F[4] * F[6]);
}
inline void cof3(const double* F, double* K) {
    K[0] = F[4] * F[8] - F[5] * F[7]; K[1] = F[5] * F[6] - F[3] * F[8]; K[2] = F[3] * F[7] - F[4] * F[6];
    K[3] = F[2] * F[7] - F[1] * F[8]; K[4] = F[0] * F[8] - F[2] * F[6]; K[5] = F[1] * F[6] - F[0] * F[7];
    K[6] = F[1] * F[5] - F[2] * F[4]; K[7] = F[2] * F[3] - F[0] * F[5]; K[8] = F[0] * F[4] - F[1] * F[3];
}
// SVD F = U diag(s) V^T with U,V rotations (Jacobi on F^T F, then U = F V / s; sign into the smallest s)
void svd3(const double* F, double* U, double* s, double* V) {
    double A[9], Ft[9]; tr(F, Ft); mm(Ft, F, A);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0);
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = std::fabs(A[1]) + std::fabs(A[2]) + std::fabs(A[5]);
        if (off <= 2e-16 * (std::fabs(A[0]) + std::fabs(A[4]) + std::fabs(A[8]))) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            double apq = A[3 * p + q];
            if (apq == 0) continue;
            double tau = (A[4 * q] - A[4 * p]) / (2 * apq);
            double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
            double c = 1 / std::sqrt(1 + t * t), sn = t * c;
            const int r = 3 - p - q;                                   // the third index
            double arp = A[3 * r + p], arq = A[3 * r + q];
            A[4 * p] -= t * apq; A[4 * q] += t * apq;
            A[3 * p + q] = A[3 * q + p] = 0;
            A[3 * r + p] = A[3 * p + r] = c * arp - sn * arq;
            A[3 * r + q] = A[3 * q + r] = sn * arp + c * arq;
            for (int k = 0; k < 3; ++k) {
                double vp = V[3 * k + p], vq = V[3 * k + q];
                V[3 * k + p] = c * vp - sn * vq; V[3 * k + q] = sn * vp + c * vq;
            }
        }
    }
    double B[9]; mm(F, V, B);
    for (int i = 0; i < 3; ++i) {
        s[i] = std::sqrt(B[i] * B[i] + B[3 + i] * B[3 + i] + B[6 + i] * B[6 + i]);
        for (int r = 0; r < 3; ++r) U[3 * r + i] = s[i] > 1e-300 ? B[3 * r + i] / s[i] : (r == i);
    }
    if (det3(V) < 0) for (int r = 0; r < 3; ++r) { V[3 * r + 2] = -V[3 * r + 2]; U[3 * r + 2] = -U[3 * r + 2]; }
    if (det3(U) < 0) {
        int k = 0; if (s[1] < s[k]) k = 1; if (s[2] < s[k]) k = 2;
        s[k] = -s[k];
        for (int r = 0; r < 3; ++r) U[3 * r + k] = -U[3 * r + k];
    }
}
inline double clamp_ref(double a) { return a >= 0 ? std::max(a, 1e-6) : std::min(a, -1e-6); }   // :184-192
// backward_svd :140-157
void backward_svd(const double* gu, const double* gs /*diag adjoint 3x3*/, const double* gv, const double* u, const double* sig3,
                  const double* v, double* out) {
    double sig[9] = {sig3[0], 0, 0, 0, sig3[1], 0, 0, 0, sig3[2]};
    double ut[9], vt[9]; tr(u, ut); tr(v, vt);
    double s2[3] = {sig3[0] * sig3[0], sig3[1] * sig3[1], sig3[2] * sig3[2]};
    double Fm[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fm[3 * i + j] = i == j ? 0.0 : 1.0 / clamp_ref(s2[j] - s2[i]);
    double a[9], b[9], gut[9], gvt[9], t1[9], t2[9];
    tr(gu, gut); tr(gv, gvt);
    mm(ut, gu, a); mm(gut, u, b);
    for (int i = 0; i < 9; ++i) t1[i] = Fm[i] * (a[i] - b[i]);
    mm(t1, sig, t1); mm(u, t1, t1); mm(t1, vt, t1);                     // u_term
    mm(vt, gv, a); mm(gvt, v, b);
    for (int i = 0; i < 9; ++i) t2[i] = Fm[i] * (a[i] - b[i]);
    mm(sig, t2, t2); mm(t2, vt, t2); mm(u, t2, t2);                     // v_term = u (sig ((F*..) vt))
    double st[9]; mm(u, gs, st); mm(st, vt, st);                        // sigma_term
    for (int i = 0; i < 9; ++i) out[i] = t1[i] + t2[i] + st[i];
}

struct Stencil { int base[3]; double fx[3], w[3][3], dw[3][3]; };
inline void stencil(const double* x, int n, Stencil& s) {                            // :215-217
    for (int d = 0; d < 3; ++d) {
        double xs = x[d] * n; int b = (int)(xs - 0.5); double fx = xs - b;
        s.base[d] = b; s.fx[d] = fx;
        s.w[0][d] = 0.5 * (1.5 - fx) * (1.5 - fx); s.w[1][d] = 0.75 - (fx - 1) * (fx - 1); s.w[2][d] = 0.5 * (fx - 0.5) * (fx - 0.5);
        s.dw[0][d] = -(1.5 - fx); s.dw[1][d] = -2 * (fx - 1); s.dw[2][d] = fx - 0.5;
    }
}
inline size_t cell(const Stencil& s, int n, int i, int j, int k) { return ((size_t)(s.base[0] + i) * n + (s.base[1] + j)) * n + (s.base[2] + k); }

// per-particle constitutive state kept between p2g and its adjoint (the reference keeps F_tmp,U,sig,V fields)
struct PState { double Ftmp[9], U[9], V[9], s[3]; };

void constitutive(const Params& P, const PState& ps, double* newF, double* stress) {   // :219-245
    const double* Ft = ps.Ftmp;
    double J = det3(Ft);
    std::memcpy(newF, Ft, 9 * sizeof(double));
    if (P.model == 0) {
        double Vt[9]; tr(ps.V, Vt);
        if (P.ptype == 0) {
            double sn[9] = {0};
            for (int d = 0; d < 3; ++d) sn[4 * d] = std::min(std::max(ps.s[d], 1 - 2e-3), 1 + 3e-3);
            mm(ps.U, sn, newF); mm(newF, Vt, newF);
        } else if (P.ptype == 2) {
            double c = std::pow(J, 1.0 / 3.0);
            for (int i = 0; i < 9; ++i) newF[i] = (i % 4 == 0) ? c : 0.0;
        }
        double R[9], D[9], nFt[9];
        mm(ps.U, Vt, R); tr(newF, nFt);
        for (int i = 0; i < 9; ++i) D[i] = newF[i] - R[i];
        mm(D, nFt, stress);
        for (int i = 0; i < 9; ++i) stress[i] = 2 * P.mu * stress[i] + ((i % 4 == 0) ? P.lam * J * (J - 1) : 0.0);
    } else {
        if (P.ptype == 2) { double sq = std::sqrt(J); double t[9] = {sq, 0, 0, 0, sq, 0, 0, 0, 1}; std::memcpy(newF, t, sizeof t); }
        double nFt[9]; tr(newF, nFt); mm(newF, nFt, stress);
        for (int i = 0; i < 9; ++i) stress[i] = P.mu * stress[i] + ((i % 4 == 0) ? P.lam * std::log(J) - P.mu : 0.0);
    }
}

// adjoint of constitutive(): G = dL/dstress, gNF = dL/dnewF  ->  gFtmp (+= ), gU, gS(diag), gV
void constitutive_grad(const Params& P, const PState& ps, const double* G, const double* gNF, double* gFt, double* gU, double* gS,
                       double* gV) {
    const double* Ft = ps.Ftmp;
    double J = det3(Ft), gJ = 0;
    for (int i = 0; i < 9; ++i) { gFt[i] = 0; gU[i] = 0; gV[i] = 0; gS[i] = 0; }
    double newF[9], stress[9];
    constitutive(P, ps, newF, stress);
    double A[9];                                                          // adjoint of newF
    std::memcpy(A, gNF, sizeof A);
    double trG = G[0] + G[4] + G[8];
    if (P.model == 0) {
        double Vt[9], R[9], X[9], Gt[9], t[9];
        tr(ps.V, Vt); mm(ps.U, Vt, R); tr(G, Gt);
        for (int i = 0; i < 9; ++i) X[i] = newF[i] - R[i];
        // L = <G, 2mu X Y^T>, Y = newF:  dX = 2mu G Y ; dY = 2mu G^T X
        double dX[9], dY[9];
        mm(G, newF, dX); mm(Gt, X, dY);
        for (int i = 0; i < 9; ++i) { dX[i] *= 2 * P.mu; dY[i] *= 2 * P.mu; A[i] += dX[i] + dY[i]; }
        double B[9]; for (int i = 0; i < 9; ++i) B[i] = -dX[i];          // adjoint of R = U V^T
        mm(B, ps.V, t); for (int i = 0; i < 9; ++i) gU[i] += t[i];        // gU += B V
        double Bt[9]; tr(B, Bt); mm(Bt, ps.U, t); for (int i = 0; i < 9; ++i) gV[i] += t[i];   // gV += B^T U
        gJ += P.lam * (2 * J - 1) * trG;
        if (P.ptype == 0) {                                               // newF = U sn V^T
            double sn[9] = {0};
            for (int d = 0; d < 3; ++d) sn[4 * d] = std::min(std::max(ps.s[d], 1 - 2e-3), 1 + 3e-3);
            double AV[9], At[9];
            mm(A, ps.V, AV); mm(AV, sn, t); for (int i = 0; i < 9; ++i) gU[i] += t[i];          // A V sn
            tr(A, At); mm(At, ps.U, t); mm(t, sn, t); for (int i = 0; i < 9; ++i) gV[i] += t[i];   // A^T U sn
            double Ut[9], M[9]; tr(ps.U, Ut); mm(Ut, A, M); mm(M, ps.V, M);
            for (int d = 0; d < 3; ++d) if (ps.s[d] > 1 - 2e-3 && ps.s[d] < 1 + 3e-3) gS[4 * d] += M[4 * d];
        } else if (P.ptype == 1) {
            for (int i = 0; i < 9; ++i) gFt[i] += A[i];
        } else {
            double c = std::pow(J, 1.0 / 3.0);
            gJ += (A[0] + A[4] + A[8]) * c / (3 * J);
        }
    } else {
        double Gs[9], t[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Gs[3 * i + j] = P.mu * (G[3 * i + j] + G[3 * j + i]);
        mm(Gs, newF, t);
        for (int i = 0; i < 9; ++i) A[i] += t[i];
        gJ += P.lam * trG / J;
        if (P.ptype == 2) gJ += (A[0] + A[4]) / (2 * std::sqrt(J));
        else for (int i = 0; i < 9; ++i) gFt[i] += A[i];
    }
    double K[9]; cof3(Ft, K);
    for (int i = 0; i < 9; ++i) gFt[i] += gJ * K[i];
}

void boundary(const Params& P, int i, int j, int k, double* v, int* mask) {             // :268-281
    const int I[3] = {i, j, k};
    *mask = 0;
    for (int d = 0; d < 3; ++d) {
        if (I[d] < 3 && v[d] < 0) { v[d] = 0; *mask |= 1 << d; }
        if (I[d] > P.n - 3 && v[d] > 0) { v[d] = 0; *mask |= 1 << d; }
    }
    if (P.sticky && j < 3) { v[0] = v[1] = v[2] = 0; *mask = 7; }
}

struct Work {
    std::vector<double> gm, gvin, gvmix, gvout, agm, agvin, agvmix, agvout, vtmp, vtgt, avtmp, avtgt;
    std::vector<PState> ps;
    void size(const Params& P) {
        size_t G = (size_t)P.n * P.n * P.n;
        gm.assign(G, 0); gvin.assign(3 * G, 0); gvmix.assign(3 * G, 0); gvout.assign(3 * G, 0);
        agm.assign(G, 0); agvin.assign(3 * G, 0); agvmix.assign(3 * G, 0); agvout.assign(3 * G, 0);
        vtmp.assign(3 * (size_t)P.N, 0); vtgt.assign(3 * (size_t)P.N, 0); avtmp.assign(3 * (size_t)P.N, 0); avtgt.assign(3 * (size_t)P.N, 0);
        ps.resize(P.N);
    }
};

inline void aadd(double* p, double v) {
#pragma omp atomic
    *p += v;
}

// forward passes up to (not including) g2p; fills W.  ext_f may be null.
void forward_grid(const Params& P, const Prim* prims, const double* pst /*P x 13*/, int f, const double* x, const double* v,
                  const double* C, const double* F, double* nF, double* ext_f, Work& W) {
    const int N = P.N, n = P.n;
    const size_t G = (size_t)n * n * n;
    W.size(P);                                                                     // clear_grid
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {                                                  // compute_F_tmp + svd
        double A[9];
        for (int i = 0; i < 9; ++i) A[i] = P.dt * C[9 * p + i] + (i % 4 == 0);
        mm(A, F + 9 * p, W.ps[p].Ftmp);
        if (P.model == 0) svd3(W.ps[p].Ftmp, W.ps[p].U, W.ps[p].s, W.ps[p].V);
    }
    const double sc = -P.dt * P.p_vol * 4 * n * (double)n;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {                                                  // p2g
        double newF[9], stress[9], aff[9];
        constitutive(P, W.ps[p], newF, stress);
        if (nF) std::memcpy(nF + 9 * p, newF, sizeof newF);
        for (int i = 0; i < 9; ++i) aff[i] = sc * stress[i] + P.p_mass * C[9 * p + i];
        Stencil s; stencil(x + 3 * p, n, s);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            double w = s.w[i][0] * s.w[j][1] * s.w[k][2];
            double dp[3] = {(i - s.fx[0]) / n, (j - s.fx[1]) / n, (k - s.fx[2]) / n};
            size_t c = cell(s, n, i, j, k);
            for (int a = 0; a < 3; ++a)
                aadd(&W.gvin[3 * c + a], w * (P.p_mass * v[3 * p + a] + aff[3 * a] * dp[0] + aff[3 * a + 1] * dp[1] + aff[3 * a + 2] * dp[2]));
            aadd(&W.gm[c], w * P.p_mass);
        }
    }
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)G; ++c) {                                           // grid_op_mixed1 / grid_op
        if (!(W.gm[c] > 1e-10)) continue;
        int k = c % n, j = (c / n) % n, i = c / ((long)n * n), mask;
        double vv[3];
        for (int a = 0; a < 3; ++a) vv[a] = W.gvin[3 * c + a] / W.gm[c] + P.dt * P.g[a];
        boundary(P, i, j, k, vv, &mask);
        for (int a = 0; a < 3; ++a) { W.gvmix[3 * c + a] = vv[a]; W.gvout[3 * c + a] = vv[a]; }
    }
    bool anyc = false;
    for (int i = 0; i < P.P; ++i) anyc |= prims[i].contact != 0;
    if (P.collision_type != 2 || !anyc) return;
    const double life = 1.0 / (P.substeps - f % P.substeps);
    std::vector<double> extl((size_t)omp_get_max_threads() * 6 * (P.P > 0 ? P.P : 1), 0.0);
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {                                                  // mixed2 + mixed3
        Stencil s; stencil(x + 3 * p, n, s);
        double vt[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            double w = s.w[i][0] * s.w[j][1] * s.w[k][2]; size_t c = cell(s, n, i, j, k);
            for (int a = 0; a < 3; ++a) vt[a] += w * W.gvmix[3 * c + a];
        }
        for (int a = 0; a < 3; ++a) W.vtmp[3 * p + a] = vt[a];
        double* acc = extl.data() + (size_t)omp_get_thread_num() * 6 * P.P;
        for (int i = 0; i < P.P; ++i) {
            if (!prims[i].contact) continue;
            double e[6];
            if (collide_mixed<double>(prims[i], pst + 13 * i, x + 3 * p, vt, P.p_mass, P.dt, life, e))
                for (int a = 0; a < 6; ++a) acc[6 * i + a] += e[a];
        }
        for (int a = 0; a < 3; ++a) W.vtgt[3 * p + a] = vt[a];
    }
    if (ext_f)
        for (int t = 0; t < omp_get_max_threads(); ++t)
            for (int a = 0; a < 6 * P.P; ++a) ext_f[a] += extl[(size_t)t * 6 * P.P + a];
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {                                                  // mixed4
        Stencil s; stencil(x + 3 * p, n, s);
        double d[3];
        for (int a = 0; a < 3; ++a) d[a] = W.vtmp[3 * p + a] - W.vtgt[3 * p + a];
        if (d[0] == 0 && d[1] == 0 && d[2] == 0) continue;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            size_t c = cell(s, n, i, j, k);
            if (!(W.gm[c] > 1e-10)) continue;
            double w = 2.0 * s.w[i][0] * s.w[j][1] * s.w[k][2];
            for (int a = 0; a < 3; ++a) aadd(&W.gvout[3 * c + a], -w * d[a]);
        }
    }
}

void g2p(const Params& P, const double* x, const Work& W, double* nx, double* nv, double* nC) {   // :299-318
    const int n = P.n;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P.N; ++p) {
        Stencil s; stencil(x + 3 * p, n, s);
        double v[3] = {0, 0, 0}, C[9] = {0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            double w = s.w[i][0] * s.w[j][1] * s.w[k][2], dp[3] = {i - s.fx[0], j - s.fx[1], k - s.fx[2]};
            const double* g = &W.gvout[3 * cell(s, n, i, j, k)];
            for (int a = 0; a < 3; ++a) { v[a] += w * g[a]; for (int b = 0; b < 3; ++b) C[3 * a + b] += 4.0 * n * w * g[a] * dp[b]; }
        }
        for (int a = 0; a < 3; ++a) { nv[3 * p + a] = v[a]; nx[3 * p + a] = x[3 * p + a] + P.dt * v[a]; }
        std::memcpy(nC + 9 * p, C, sizeof C);
    }
}

Work g_work;

}  // namespace

extern "C" {

struct mc_params { int N, n_grid, substeps, ptype, model, P, sticky, collision_type; double dt, mu, lam, p_vol, p_mass, g[3]; };
struct mc_prim { const double* sdf; const double* normal; int res[3]; int contact; double lower[3], upper[3], sdf_dx, friction, softness; };

static Params to_params(const mc_params* m) {
    Params P;
    P.N = m->N; P.n = m->n_grid; P.substeps = m->substeps; P.ptype = m->ptype; P.model = m->model; P.P = m->P; P.sticky = m->sticky;
    P.collision_type = m->collision_type; P.dt = m->dt; P.mu = m->mu; P.lam = m->lam; P.p_vol = m->p_vol; P.p_mass = m->p_mass;
    for (int i = 0; i < 3; ++i) P.g[i] = m->g[i];
    return P;
}
static std::vector<Prim> to_prims(const mc_params* m, const mc_prim* pr) {
    std::vector<Prim> out(m->P);
    for (int i = 0; i < m->P; ++i) {
        out[i].sdf = pr[i].sdf; out[i].normal = pr[i].normal; out[i].contact = pr[i].contact;
        out[i].inv_dx = 1.0 / pr[i].sdf_dx; out[i].friction = pr[i].friction; out[i].softness = pr[i].softness;
        for (int d = 0; d < 3; ++d) { out[i].res[d] = pr[i].res[d]; out[i].lower[d] = pr[i].lower[d]; out[i].upper[d] = pr[i].upper[d]; }
    }
    return out;
}

int mc_threads(void) { return omp_get_max_threads(); }
void mc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }   // bench.py's single-thread figure

// One forward substep (substep :320-337).  Arrays AOS f64: x,v (N,3); C,F (N,3,3).  ext_f (P,6) is ACCUMULATED.
void mc_substep(const mc_params* m, const mc_prim* pr, const double* pst, int f, const double* x, const double* v, const double* C,
                const double* F, double* nx, double* nv, double* nC, double* nF, double* ext_f) {
    Params P = to_params(m);
    std::vector<Prim> prims = to_prims(m, pr);
    forward_grid(P, prims.data(), pst, f, x, v, C, F, nF, ext_f, g_work);
    g2p(P, x, g_work, nx, nv, nC);
}

// Adjoint of one substep (substep_grad :339-378): recompute, then reverse.  g*1: adjoints of frame f+1;
// g*0: adjoints of frame f, ACCUMULATED (+=) like the reference's fields; gpst (P,13) accumulated; egrad (P,6) seeds.
void mc_substep_grad(const mc_params* m, const mc_prim* pr, const double* pst, int f, const double* x, const double* v, const double* C,
                     const double* F, const double* gx1, const double* gv1, const double* gC1, const double* gF1, const double* egrad,
                     double* gx0, double* gv0, double* gC0, double* gF0, double* gpst) {
    Params P = to_params(m);
    std::vector<Prim> prims = to_prims(m, pr);
    Work& W = g_work;
    const int N = P.N, n = P.n;
    const size_t G = (size_t)n * n * n;
    forward_grid(P, prims.data(), pst, f, x, v, C, F, nullptr, nullptr, W);
    // ---- g2p.grad
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {
        Stencil s; stencil(x + 3 * p, n, s);
        double gnv[3], gfx[3] = {0, 0, 0}, gw[3][3] = {{0}};
        for (int a = 0; a < 3; ++a) { gnv[a] = gv1[3 * p + a] + P.dt * gx1[3 * p + a]; gx0[3 * p + a] += gx1[3 * p + a]; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            double w = s.w[i][0] * s.w[j][1] * s.w[k][2], dp[3] = {i - s.fx[0], j - s.fx[1], k - s.fx[2]};
            size_t c = cell(s, n, i, j, k);
            double gwn = 0, gdp[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                double t = gnv[a];
                for (int b = 0; b < 3; ++b) t += 4.0 * n * gC1[9 * p + 3 * a + b] * dp[b];
                aadd(&W.agvout[3 * c + a], w * t);
                gwn += W.gvout[3 * c + a] * t;
                for (int b = 0; b < 3; ++b) gdp[b] += 4.0 * n * W.gvout[3 * c + a] * gC1[9 * p + 3 * a + b];
            }
            gw[i][0] += gwn * s.w[j][1] * s.w[k][2]; gw[j][1] += gwn * s.w[i][0] * s.w[k][2]; gw[k][2] += gwn * s.w[i][0] * s.w[j][1];
            for (int b = 0; b < 3; ++b) gfx[b] -= w * gdp[b];
        }
        for (int d = 0; d < 3; ++d) { gfx[d] += gw[0][d] * s.dw[0][d] + gw[1][d] * s.dw[1][d] + gw[2][d] * s.dw[2][d]; gx0[3 * p + d] += n * gfx[d]; }
    }
    bool anyc = false;
    for (int i = 0; i < P.P; ++i) anyc |= prims[i].contact != 0;
    if (P.collision_type == 2 && anyc) {
        const double life = 1.0 / (P.substeps - f % P.substeps);
        std::vector<double> gl((size_t)omp_get_max_threads() * 13 * P.P, 0.0);
        // ---- mixed4.grad, mixed3.grad, mixed2.grad (fused per particle; v_tmp/v_tgt adjoints are private)
#pragma omp parallel for schedule(static)
        for (int p = 0; p < N; ++p) {
            Stencil s; stencil(x + 3 * p, n, s);
            double d[3], gd[3] = {0, 0, 0}, gw[3][3] = {{0}};
            for (int a = 0; a < 3; ++a) d[a] = W.vtmp[3 * p + a] - W.vtgt[3 * p + a];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
                size_t c = cell(s, n, i, j, k);
                if (!(W.gm[c] > 1e-10)) continue;
                double w = s.w[i][0] * s.w[j][1] * s.w[k][2], dg = 0;
                for (int a = 0; a < 3; ++a) { gd[a] -= 2.0 * w * W.agvout[3 * c + a]; dg += d[a] * W.agvout[3 * c + a]; }
                double gwn = -2.0 * dg;
                gw[i][0] += gwn * s.w[j][1] * s.w[k][2]; gw[j][1] += gwn * s.w[i][0] * s.w[k][2]; gw[k][2] += gwn * s.w[i][0] * s.w[j][1];
            }
            // mixed3.grad: chain of primitives in reverse, velocities entering each primitive replayed forward
            double g[3] = {-gd[0], -gd[1], -gd[2]}, gpos[3] = {0, 0, 0};
            double vin[8][3], cur[3] = {W.vtmp[3 * p], W.vtmp[3 * p + 1], W.vtmp[3 * p + 2]}, e6[6];
            bool act[8];
            for (int i = 0; i < P.P; ++i) {
                for (int a = 0; a < 3; ++a) vin[i][a] = cur[a];
                act[i] = prims[i].contact && collide_mixed<double>(prims[i], pst + 13 * i, x + 3 * p, cur, P.p_mass, P.dt, life, e6);
            }
            double* gacc = gl.data() + (size_t)omp_get_thread_num() * 13 * P.P;
            for (int i = P.P - 1; i >= 0; --i) {
                if (!act[i]) continue;
                double out[19];
                for (int dir = 0; dir < 19; ++dir) {
                    Dual xs[3], vs[3], st[13], ex[6];
                    for (int a = 0; a < 3; ++a) { xs[a] = Dual(x[3 * p + a], dir == a); vs[a] = Dual(vin[i][a], dir == 3 + a); }
                    for (int a = 0; a < 13; ++a) st[a] = Dual(pst[13 * i + a], dir == 6 + a);
                    collide_mixed<Dual>(prims[i], st, xs, vs, P.p_mass, P.dt, life, ex);
                    double sacc = 0;
                    for (int a = 0; a < 3; ++a) sacc += g[a] * vs[a].d;
                    if (egrad) for (int a = 0; a < 6; ++a) sacc += egrad[6 * i + a] * ex[a].d;
                    out[dir] = sacc;
                }
                for (int a = 0; a < 3; ++a) { gpos[a] += out[a]; g[a] = out[3 + a]; }
                for (int a = 0; a < 13; ++a) gacc[13 * i + a] += out[6 + a];
            }
            double gvt[3] = {gd[0] + g[0], gd[1] + g[1], gd[2] + g[2]};
            if (gvt[0] != 0 || gvt[1] != 0 || gvt[2] != 0 || gd[0] != 0 || gd[1] != 0 || gd[2] != 0) {
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {   // mixed2.grad
                    double w = s.w[i][0] * s.w[j][1] * s.w[k][2], gwn = 0;
                    size_t c = cell(s, n, i, j, k);
                    for (int a = 0; a < 3; ++a) { aadd(&W.agvmix[3 * c + a], w * gvt[a]); gwn += W.gvmix[3 * c + a] * gvt[a]; }
                    gw[i][0] += gwn * s.w[j][1] * s.w[k][2]; gw[j][1] += gwn * s.w[i][0] * s.w[k][2]; gw[k][2] += gwn * s.w[i][0] * s.w[j][1];
                }
            }
            for (int dd = 0; dd < 3; ++dd)
                gx0[3 * p + dd] += gpos[dd] + n * (gw[0][dd] * s.dw[0][dd] + gw[1][dd] * s.dw[1][dd] + gw[2][dd] * s.dw[2][dd]);
        }
        if (gpst)
            for (int t = 0; t < omp_get_max_threads(); ++t)
                for (int a = 0; a < 13 * P.P; ++a) gpst[a] += gl[(size_t)t * 13 * P.P + a];
    }
    // ---- grid_op_mixed1.grad
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)G; ++c) {
        if (!(W.gm[c] > 1e-10)) continue;
        int k = c % n, j = (c / n) % n, i = c / ((long)n * n), mask;
        double vv[3], g[3], gmm = 0;
        for (int a = 0; a < 3; ++a) { vv[a] = W.gvin[3 * c + a] / W.gm[c] + P.dt * P.g[a]; g[a] = W.agvout[3 * c + a] + W.agvmix[3 * c + a]; }
        boundary(P, i, j, k, vv, &mask);
        for (int a = 0; a < 3; ++a) {
            if (mask & (1 << a)) g[a] = 0;
            W.agvin[3 * c + a] = g[a] / W.gm[c];
            gmm -= W.gvin[3 * c + a] * g[a];
        }
        W.agm[c] = gmm / (W.gm[c] * W.gm[c]);
    }
    // ---- p2g.grad + svd_grad + compute_F_tmp.grad
    const double sc = -P.dt * P.p_vol * 4 * n * (double)n;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < N; ++p) {
        const PState& ps = W.ps[p];
        double newF[9], stress[9], aff[9];
        constitutive(P, ps, newF, stress);
        for (int i = 0; i < 9; ++i) aff[i] = sc * stress[i] + P.p_mass * C[9 * p + i];
        Stencil s; stencil(x + 3 * p, n, s);
        double gvp[3] = {0, 0, 0}, gaff[9] = {0}, gfx[3] = {0, 0, 0}, gw[3][3] = {{0}};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) {
            double w = s.w[i][0] * s.w[j][1] * s.w[k][2], dp[3] = {(i - s.fx[0]) / n, (j - s.fx[1]) / n, (k - s.fx[2]) / n};
            size_t c = cell(s, n, i, j, k);
            double gwn = W.agm[c] * P.p_mass, gdp[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                double gv = W.agvin[3 * c + a];
                gwn += gv * (P.p_mass * v[3 * p + a] + aff[3 * a] * dp[0] + aff[3 * a + 1] * dp[1] + aff[3 * a + 2] * dp[2]);
                gvp[a] += w * gv;
                for (int b = 0; b < 3; ++b) { gaff[3 * a + b] += w * gv * dp[b]; gdp[b] += aff[3 * a + b] * gv; }
            }
            gw[i][0] += gwn * s.w[j][1] * s.w[k][2]; gw[j][1] += gwn * s.w[i][0] * s.w[k][2]; gw[k][2] += gwn * s.w[i][0] * s.w[j][1];
            for (int b = 0; b < 3; ++b) gfx[b] -= w * gdp[b] / n;
        }
        for (int d = 0; d < 3; ++d) {
            gfx[d] += gw[0][d] * s.dw[0][d] + gw[1][d] * s.dw[1][d] + gw[2][d] * s.dw[2][d];
            gx0[3 * p + d] += n * gfx[d];
            gv0[3 * p + d] += P.p_mass * gvp[d];
        }
        double Gs[9], gFt[9], gU[9], gS[9], gV[9];
        for (int i = 0; i < 9; ++i) Gs[i] = sc * gaff[i];
        constitutive_grad(P, ps, Gs, gF1 + 9 * p, gFt, gU, gS, gV);
        if (P.model == 0) {                                                       // svd_grad :135-138
            double t[9];
            backward_svd(gU, gS, gV, ps.U, ps.s, ps.V, t);
            for (int i = 0; i < 9; ++i) gFt[i] += t[i];
        }
        // compute_F_tmp.grad: F_tmp = (I + dt C) F
        double Ft_[9], A[9], At[9], t1[9], t2[9];
        tr(F + 9 * p, Ft_); mm(gFt, Ft_, t1);
        for (int i = 0; i < 9; ++i) A[i] = P.dt * C[9 * p + i] + (i % 4 == 0);
        tr(A, At); mm(At, gFt, t2);
        for (int i = 0; i < 9; ++i) { gC0[9 * p + i] += P.dt * t1[i] + P.p_mass * gaff[i]; gF0[9 * p + i] += t2[i]; }
    }
}

}  // extern "C"
