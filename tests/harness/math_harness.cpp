// Test-only host build of softmac_amd/csrc/smac_math.hpp (g++).  Lets pytest compare the
// per-particle device arithmetic with the torch oracle on the CPU.  Never linked into the product.
#include <vector>
#include "../../softmac_amd/csrc/smac_math.hpp"

using namespace smac;

template <class R>
static void constitutive_run(int n, int ptype, int model, double mu, double lam, const double* Et, const double* G,
                             const double* gFn, double* En, double* stress, double* gEt, int plast = PLAST_CLIP, double yield_c = 0.0) {
    Material<R> M{ptype, model, (R)mu, (R)lam, plast, (R)yield_c};
    for (int p = 0; p < n; ++p) {
        R e[9], g[9], gf[9], en[9], st[9], ge[9];
        for (int i = 0; i < 9; ++i) { e[i] = (R)Et[9 * p + i]; g[i] = (R)G[9 * p + i]; gf[i] = (R)gFn[9 * p + i]; }
        ConstState<R> cs;
        constitutive_fwd(M, e, en, st, cs);
        constitutive_bwd(M, e, cs, g, gf, ge);
        for (int i = 0; i < 9; ++i) { En[9 * p + i] = en[i]; stress[9 * p + i] = st[i]; gEt[9 * p + i] = ge[i]; }
    }
}

template <class R>
static void svd_run(int n, const double* E, double* U, double* e, double* V) {
    for (int p = 0; p < n; ++p) {
        R a[9], u[9], v[9], s[3];
        for (int i = 0; i < 9; ++i) a[i] = (R)E[9 * p + i];
        svd_I_plus_E(a, u, s, v);
        for (int i = 0; i < 9; ++i) { U[9 * p + i] = u[i]; V[9 * p + i] = v[i]; }
        for (int i = 0; i < 3; ++i) e[3 * p + i] = s[i];
    }
}

template <class R>
static void collide_run(int n, const double* sdf, const double* normal, const int* res, const double* lower,
                        const double* upper, double sdf_dx, double friction, double softness, const double* st13,
                        const double* pos, const double* vel, double p_mass, double dt, double life,
                        const double* g_v, const double* g_ext, double* out_v, double* out_ext, int* active,
                        double* g_pos, double* g_vin, double* g_state) {
    long cells = (long)res[0] * res[1] * res[2];
    R* ts = new R[cells];
    R* tn = new R[cells * 3];
    for (long i = 0; i < cells; ++i) ts[i] = (R)sdf[i];
    for (long i = 0; i < cells * 3; ++i) tn[i] = (R)normal[i];
    PrimTable<R> T;
    T.sdf = ts; T.normal = tn;
    for (int i = 0; i < 3; ++i) { T.res[i] = res[i]; T.lower[i] = (R)lower[i]; T.upper[i] = (R)upper[i]; }
    T.inv_dx = (R)(1.0 / sdf_dx); T.friction = (R)friction; T.softness = (R)softness; T.contact = 1;
    R st[13];
    for (int i = 0; i < 13; ++i) st[i] = (R)st13[i];
    for (int i = 0; i < 13; ++i) g_state[i] = 0;
    for (int p = 0; p < n; ++p) {
        R x[3], v[3], vin[3], ext[6] = {0, 0, 0, 0, 0, 0}, gv[3], ge[6], gp[3] = {0, 0, 0}, gvi[3], gs[13];
        for (int i = 0; i < 3; ++i) { x[i] = (R)pos[3 * p + i]; v[i] = vin[i] = (R)vel[3 * p + i]; gv[i] = (R)g_v[3 * p + i]; }
        for (int i = 0; i < 6; ++i) ge[i] = (R)g_ext[i];
        for (int i = 0; i < 13; ++i) gs[i] = 0;
        bool act = collide_mixed(T, st, x, v, (R)p_mass, (R)dt, (R)life, ext);
        active[p] = act;
        for (int i = 0; i < 3; ++i) out_v[3 * p + i] = v[i];
        for (int i = 0; i < 6; ++i) out_ext[6 * p + i] = ext[i];
        for (int i = 0; i < 3; ++i) gvi[i] = gv[i];   // identity outside the band
        collide_mixed_adjoint(T, st, x, vin, (R)p_mass, (R)dt, (R)life, gv, ge, gp, gvi, gs);
        for (int i = 0; i < 3; ++i) { g_pos[3 * p + i] = gp[i]; g_vin[3 * p + i] = gvi[i]; }
        for (int i = 0; i < 13; ++i) g_state[i] += gs[i];
    }
    delete[] ts;
    delete[] tn;
}

// the two-width chain of float32 mode (collide_mixed_hybrid): f64 distance, f32 everything else; positions and primitive state in f64, velocities in f32
static void collide_hybrid_run(int n, const double* sdf, const double* normal, const int* res, const double* lower,
                               const double* upper, double sdf_dx, double friction, double softness, const double* st13,
                               const double* pos, const double* vel, double p_mass, double dt, double life,
                               const double* g_v, const double* g_ext, double* out_v, double* out_ext, int* active,
                               double* g_pos, double* g_vin, double* g_state) {
    long cells = (long)res[0] * res[1] * res[2];
    std::vector<float> ts32(cells), tn32(cells * 3);
    for (long i = 0; i < cells; ++i) ts32[i] = (float)sdf[i];
    for (long i = 0; i < cells * 3; ++i) tn32[i] = (float)normal[i];
    PrimTable<double> T64;
    PrimTable<float> T32;
    T64.sdf = sdf; T64.normal = normal; T32.sdf = ts32.data(); T32.normal = tn32.data();
    for (int i = 0; i < 3; ++i) { T64.res[i] = T32.res[i] = res[i]; T64.lower[i] = lower[i]; T64.upper[i] = upper[i]; T32.lower[i] = (float)lower[i]; T32.upper[i] = (float)upper[i]; }
    T64.inv_dx = 1.0 / sdf_dx; T64.friction = friction; T64.softness = softness; T64.contact = 1;
    T32.inv_dx = (float)(1.0 / sdf_dx); T32.friction = (float)friction; T32.softness = (float)softness; T32.contact = 1;
    for (int i = 0; i < 13; ++i) g_state[i] = 0;
    for (int p = 0; p < n; ++p) {
        double x[3];
        float v[3], ext[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; ++i) { x[i] = pos[3 * p + i]; v[i] = (float)vel[3 * p + i]; }
        const float vin[3] = {v[0], v[1], v[2]};
        const bool act = collide_mixed_hybrid<double, float>(T64, T32, st13, x, v, (float)p_mass, dt, life, ext);
        active[p] = act;
        for (int i = 0; i < 3; ++i) out_v[3 * p + i] = v[i];
        for (int i = 0; i < 6; ++i) out_ext[6 * p + i] = ext[i];
        for (int i = 0; i < 3; ++i) { g_pos[3 * p + i] = 0; g_vin[3 * p + i] = g_v[3 * p + i]; }     // identity outside the band
        if (!act) continue;
        for (int dir = 0; dir < 19; ++dir) {
            Dual<double> xs[3], ss[13];
            Dual<float> vs[3], es[6];
            for (int i = 0; i < 3; ++i) { xs[i] = Dual<double>(x[i], dir == i ? 1.0 : 0.0); vs[i] = Dual<float>(vin[i], dir == 3 + i ? 1.f : 0.f); }
            for (int i = 0; i < 13; ++i) ss[i] = Dual<double>(st13[i], dir == 6 + i ? 1.0 : 0.0);
            collide_mixed_hybrid<Dual<double>, Dual<float>>(T64, T32, ss, xs, vs, (float)p_mass, dt, life, es);
            double acc = 0;
            for (int i = 0; i < 3; ++i) acc += g_v[3 * p + i] * (double)vs[i].d;
            for (int i = 0; i < 6; ++i) acc += g_ext[i] * (double)es[i].d;
            if (dir < 3) g_pos[3 * p + dir] = acc;
            else if (dir < 6) g_vin[3 * p + dir - 3] = acc;
            else g_state[dir - 6] += acc;
        }
    }
}

// collide_particle / collide (grid) forward + forward-mode adjoints, one particle / node per entry
template <class R>
static void collide_other_run(int kind, int n, const double* sdf, const double* normal, const int* res, const double* lower,
                              const double* upper, double sdf_dx, double friction, double softness, const double* st13,
                              const double* pos, const double* vel, const double* mass, double dt, const double* g_out,
                              const double* g_ext, double* out3, double* out_ext, int* active, double* g_in /* n x 20 */) {
    long cells = (long)res[0] * res[1] * res[2];
    std::vector<R> ts(cells), tn(cells * 3);
    for (long i = 0; i < cells; ++i) ts[i] = (R)sdf[i];
    for (long i = 0; i < cells * 3; ++i) tn[i] = (R)normal[i];
    PrimTable<R> T;
    T.sdf = ts.data(); T.normal = tn.data();
    for (int i = 0; i < 3; ++i) { T.res[i] = res[i]; T.lower[i] = (R)lower[i]; T.upper[i] = (R)upper[i]; }
    T.inv_dx = (R)(1.0 / sdf_dx); T.friction = (R)friction; T.softness = (R)softness; T.contact = 1;
    for (int p = 0; p < n; ++p) {
        R st[13], x[3], v[3], o[3] = {0, 0, 0}, ext[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 13; ++i) st[i] = (R)st13[i];
        for (int i = 0; i < 3; ++i) { x[i] = (R)pos[3 * p + i]; v[i] = (R)vel[3 * p + i]; }
        bool act;
        if (kind == 1) act = collide_particle(T, st, x, v, (R)dt, o, ext);
        else { for (int i = 0; i < 3; ++i) o[i] = v[i]; act = collide_grid(T, st, x, o, (R)mass[p], (R)dt, ext); }
        active[p] = act;
        for (int i = 0; i < 3; ++i) out3[3 * p + i] = o[i];
        for (int i = 0; i < 6; ++i) out_ext[6 * p + i] = ext[i];
        // directions: 0-2 pos (particle contact only), 3-5 v, 6 mass (grid contact only), 7-19 state
        for (int dir = 0; dir < 20; ++dir) {
            double acc = 0;
            if (act) {
                Dual<R> xs[3], vs[3], ss[13], os[3], es[6];
                for (int i = 0; i < 3; ++i) { xs[i] = Dual<R>(x[i], (kind == 1 && dir == i) ? R(1) : R(0)); vs[i] = Dual<R>(v[i], dir == 3 + i ? R(1) : R(0)); }
                for (int i = 0; i < 13; ++i) ss[i] = Dual<R>(st[i], dir == 7 + i ? R(1) : R(0));
                if (kind == 1) collide_particle(T, ss, xs, vs, (R)dt, os, es);
                else { for (int i = 0; i < 3; ++i) os[i] = vs[i]; collide_grid(T, ss, x, os, Dual<R>((R)mass[p], dir == 6 ? R(1) : R(0)), (R)dt, es); }
                for (int i = 0; i < 3; ++i) acc += g_out[3 * p + i] * os[i].d;
                for (int i = 0; i < 6; ++i) acc += g_ext[i] * es[i].d;
            } else if (kind == 0 && dir >= 3 && dir < 6) acc = g_out[3 * p + dir - 3];
            g_in[20 * p + dir] = acc;
        }
    }
}

template <class R> static void fk_run(const double* s13, double dt, double* out7) {
    R s[13], o[7];
    for (int i = 0; i < 13; ++i) s[i] = (R)s13[i];
    forward_kinematics(s, (R)dt, o);
    for (int i = 0; i < 7; ++i) out7[i] = o[i];
}

extern "C" {
void h_constitutive(int prec, int n, int ptype, int model, double mu, double lam, const double* Et, const double* G,
                    const double* gFn, double* En, double* stress, double* gEt) {
    if (prec == 64) constitutive_run<double>(n, ptype, model, mu, lam, Et, G, gFn, En, stress, gEt);
    else constitutive_run<float>(n, ptype, model, mu, lam, Et, G, gFn, En, stress, gEt);
}
void h_constitutive_von_mises(int prec, int n, double mu, double lam, double yield_c, const double* Et, const double* G, const double* gFn,
                              double* En, double* stress, double* gEt) {
    if (prec == 64) constitutive_run<double>(n, 0, 0, mu, lam, Et, G, gFn, En, stress, gEt, PLAST_VON_MISES, yield_c);
    else constitutive_run<float>(n, 0, 0, mu, lam, Et, G, gFn, En, stress, gEt, PLAST_VON_MISES, yield_c);
}
void h_svd(int prec, int n, const double* E, double* U, double* e, double* V) {
    if (prec == 64) svd_run<double>(n, E, U, e, V); else svd_run<float>(n, E, U, e, V);
}
void h_collide_mixed(int prec, int n, const double* sdf, const double* normal, const int* res, const double* lower,
                     const double* upper, double sdf_dx, double friction, double softness, const double* st13,
                     const double* pos, const double* vel, double p_mass, double dt, double life, const double* g_v,
                     const double* g_ext, double* out_v, double* out_ext, int* active, double* g_pos, double* g_vin,
                     double* g_state) {
    if (prec == 48)
        collide_hybrid_run(n, sdf, normal, res, lower, upper, sdf_dx, friction, softness, st13, pos, vel, p_mass, dt,
                           life, g_v, g_ext, out_v, out_ext, active, g_pos, g_vin, g_state);
    else if (prec == 64)
        collide_run<double>(n, sdf, normal, res, lower, upper, sdf_dx, friction, softness, st13, pos, vel, p_mass, dt,
                            life, g_v, g_ext, out_v, out_ext, active, g_pos, g_vin, g_state);
    else
        collide_run<float>(n, sdf, normal, res, lower, upper, sdf_dx, friction, softness, st13, pos, vel, p_mass, dt,
                           life, g_v, g_ext, out_v, out_ext, active, g_pos, g_vin, g_state);
}
void h_collide_other(int prec, int kind, int n, const double* sdf, const double* normal, const int* res, const double* lower,
                     const double* upper, double sdf_dx, double friction, double softness, const double* st13, const double* pos,
                     const double* vel, const double* mass, double dt, const double* g_out, const double* g_ext, double* out3,
                     double* out_ext, int* active, double* g_in) {
    if (prec == 64) collide_other_run<double>(kind, n, sdf, normal, res, lower, upper, sdf_dx, friction, softness, st13, pos, vel, mass, dt, g_out, g_ext, out3, out_ext, active, g_in);
    else collide_other_run<float>(kind, n, sdf, normal, res, lower, upper, sdf_dx, friction, softness, st13, pos, vel, mass, dt, g_out, g_ext, out3, out_ext, active, g_in);
}
void h_forward_kinematics(int prec, const double* s13, double dt, double* out7) {
    if (prec == 64) fk_run<double>(s13, dt, out7); else fk_run<float>(s13, dt, out7);
}
}
