// Soft <-> cloth contact: geometry of one triangle of the cloth primitive and the two contact models built on it,
// written for a generic scalar S (double, or Dual<double> for the forward-mode adjoint passes).  The sheet's vertices move
// kinematically: positions / velocities per frame are inputs, and receive adjoints.
//   reference: soft_cloth/engine/primitive/primitive_cloth.py (cited by line below)
// Everything here runs in double whatever the storage type of the particle kernels is: the push-out of a penetrated particle
// divides a distance by dt, like the SDF primitives' forecast contact.
#pragma once
#include "smac_math.hpp"

namespace smac {

struct ClothParams {
    double friction, softness, force_scale, scale;   // friction[None], softness[None], cloth_force_scale[None], mpm_scale (:65-68, :41)
    int sticky;                                       // cfg.sticky (:69, :376)
};

// device-side view of the cloth primitive (Primitive_Cloth :26-69 + the per-particle contact arrays of soft_cloth's
// MPMSimulator :81-83).  Vertex data is f64 and physical (units of mpm_scale); per-particle arrays are indexed by ORIGINAL id.
struct ClothDev {
    int V, Fc, n_neighbors, present;
    const int* faces;                 // [Fc][3]
    const int* nbr;                   // [Fc][n_neighbors]      neighbor_faces
    const signed char* nbr_dir;       // [Fc][n_neighbors]      neighbor_faces_direction
    double *pos, *vel;                // [max_frames][V][3]
    double *pos_grad, *vel_grad;      // same shape (nullptr without gradients)
    double *ext_f, *ext_f_grad;       // [V][3]
    int* contact_id;                  // [max_frames][N]
    signed char* penetration;         // [max_frames][N]
    int* contact_before;              // [N]                    contact_id_before_cloth
    int n_ids;                        // N the per-particle arrays were sized for
    ClothParams par;
};

template <class S> SMAC_HD S cl_length(const S* x) { return sqrt_(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + 1e-14); }   // :18-20
template <class S> SMAC_HD void cl_normalize(S* n) {                                                                    // :22-24
    const S l = cl_length(n);
    n[0] = n[0] / l; n[1] = n[1] / l; n[2] = n[2] / l;
}

// closest_point_on_edge :83-96
template <class S> SMAC_HD void cl_closest_on_edge(const S* p, const S* x0, const S* x1, S* out) {
    S v[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]};
    S w[3] = {p[0] - x0[0], p[1] - x0[1], p[2] - x0[2]};
    const S c1 = dot3(w, v), c2 = dot3(v, v);
    if (val(c1) >= val(c2)) { out[0] = x1[0]; out[1] = x1[1]; out[2] = x1[2]; }
    else if (val(c1) > 0.0) {
        const S t = c1 / c2;
        out[0] = x0[0] + v[0] * t; out[1] = x0[1] + v[1] * t; out[2] = x0[2] + v[2] * t;
    } else { out[0] = x0[0]; out[1] = x0[1]; out[2] = x0[2]; }
}

// barycentric_coordinate :98-113 (p must lie in the plane of the triangle)
template <class S> SMAC_HD void cl_barycentric(const S* p, const S* x0, const S* x1, const S* x2, S* w) {
    S A[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]};
    S B[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    S C[3] = {p[0] - x0[0], p[1] - x0[1], p[2] - x0[2]};
    const S dxy = A[0] * B[1] - A[1] * B[0];
    const double a = val(dxy);
    if ((a < 0 ? -a : a) < 1e-10) {
        w[0] = (C[0] * B[2] - C[2] * B[0]) / (A[0] * B[2] - A[2] * B[0]);
        w[1] = (C[0] * A[2] - C[2] * A[0]) / (B[0] * A[2] - B[2] * A[0]);
    } else {
        w[0] = (C[0] * B[1] - C[1] * B[0]) / dxy;
        w[1] = (C[0] * A[1] - C[1] * A[0]) / (B[0] * A[1] - B[1] * A[0]);
    }
    w[2] = 1.0 - w[0] - w[1];
}

// shared body of distance_function :120-135 and sdf_and_normal :142-158: distance to the plane inside the triangle, else to the
// nearest edge, with the direction it is measured along
template <class S> SMAC_HD S cl_plane_or_edge(const S* p, const S* x0, const S* x1, const S* x2, S* n) {
    S e1[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]}, e2[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    cross3(e1, e2, n);
    cl_normalize(n);
    S r[3] = {p[0] - x0[0], p[1] - x0[1], p[2] - x0[2]};
    S d = dot3(n, r);
    S q[3] = {p[0] - d * n[0], p[1] - d * n[1], p[2] - d * n[2]}, w[3];
    cl_barycentric(q, x0, x1, x2, w);
    if (!(val(w[0]) >= 0.0 && val(w[1]) >= 0.0 && val(w[2]) >= 0.0)) {       // point_in_triangle :115-118
        d = S(1e6);
        const S* xs[3] = {x0, x1, x2};
        for (int i = 0; i < 3; ++i) {
            S pt[3];
            cl_closest_on_edge(p, xs[i], xs[(i + 1) % 3], pt);
            S df[3] = {p[0] - pt[0], p[1] - pt[1], p[2] - pt[2]};
            const S dl = cl_length(df);
            if (val(dl) < val(d)) {
                d = dl;
                n[0] = df[0] / dl; n[1] = df[1] / dl; n[2] = df[2] / dl;       // normalize(p_pos - point) = diff / length(diff)
            }
        }
    }
    return d;
}
SMAC_HD double cl_distance(const double* p, const double* x0, const double* x1, const double* x2) {     // distance_function :120-140
    double n[3];
    const double d = cl_plane_or_edge(p, x0, x1, x2, n);
    return d < 0 ? -d : d;
}
// sdf_and_normal :142-164
template <class S> SMAC_HD S cl_sdf_and_normal(const S* p, int penetrated, const S* x0, const S* x1, const S* x2, S* n) {
    S d = cl_plane_or_edge(p, x0, x1, x2, n);
    if ((penetrated == 0) == (val(d) < 0.0)) {                                 // :160-162
        d = -d;
        n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2];
    }
    return d;
}
// in_bounding_box :166-179
SMAC_HD bool cl_in_bbox(const double* p, const double* x0, const double* x1, const double* x2, double threshold) {
    for (int i = 0; i < 3; ++i) {
        const double lo = min_(x0[i], min_(x1[i], x2[i])) - threshold, hi = max_(x0[i], max_(x1[i], x2[i])) + threshold;
        if (p[i] <= lo || p[i] >= hi) return false;
    }
    return true;
}
// check_side :189-196 (the normal is not normalised)
SMAC_HD bool cl_check_side(const double* p, const double* x0, const double* x1, const double* x2) {
    const double e1[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]}, e2[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
    double n[3];
    cross3(e1, e2, n);
    return n[0] * (p[0] - x0[0]) + n[1] * (p[1] - x0[1]) + n[2] * (p[2] - x0[2]) > 0.0;
}

// collide_mixed :233-280.  xv = the face's three vertex positions, vv = their velocities.  Returns true inside the band; then v_io is the
// target velocity, cf the force on the cloth at the contact point and w its barycentric split over the three vertices (:274-278).
template <class S>
SMAC_HD bool cloth_collide_mixed(const ClothParams& P, const S (*xv)[3], const S (*vv)[3], const S* p_pos, S* v_io, double p_mass, double dt,
                                 double life, int penetrated, S* cf, S* w) {
    S D[3];
    const S dist = cl_sdf_and_normal(p_pos, penetrated, xv[0], xv[1], xv[2], D);
    if (!(val(dist) <= 5e-3 * P.scale)) return false;                           // :236-237
    S q[3] = {p_pos[0] - D[0] * dist, p_pos[1] - D[1] * dist, p_pos[2] - D[2] * dist};
    cl_barycentric(q, xv[0], xv[1], xv[2], w);                                  // :242
    S cv[3], in[3], pv[3];
    for (int c = 0; c < 3; ++c) {
        cv[c] = w[0] * vv[0][c] + w[1] * vv[1][c] + w[2] * vv[2][c];            // :244-245
        in[c] = v_io[c] - cv[c];                                                // :247
        pv[c] = v_io[c];
    }
    const S nc = dot3(in, D);
    if (!P.sticky) {                                                            // :250-262
        if (val(nc) < 0.0) {
            S t[3] = {in[0] - nc * D[0], in[1] - nc * D[1], in[2] - nc * D[2]};
            const S tt = dot3(t, t);
            const S tn = sqrt_(tt + 1e-14);
            const S sc = maxc(tn + nc * P.friction, 0.0) / tn;                  // :254
            const double flag = std::sqrt(val(tt)) > 1e-30 ? 1.0 : 0.0;         // :255
            for (int c = 0; c < 3; ++c) t[c] = (t[c] * sc) * flag + t[c] * (1.0 - flag);
            if (val(dist) > 0.0) {
                const S infl = minc(exp_(-dist * P.softness), 1.0);             // :261
                for (int c = 0; c < 3; ++c) pv[c] = cv[c] + in[c] * (1.0 - infl) + t[c] * infl;
            } else {
                for (int c = 0; c < 3; ++c) pv[c] = cv[c] + t[c];               // :258
            }
        }
    } else {                                                                    // :263-268
        if (val(dist) > 0.0) {
            const S infl = minc(exp_(-dist * P.softness), 1.0);
            for (int c = 0; c < 3; ++c) pv[c] = cv[c] + in[c] * (1.0 - infl);
        } else {
            for (int c = 0; c < 3; ++c) pv[c] = cv[c];
        }
    }
    if (val(dist) < 0.0) {                                                      // :271-272
        const S k = -(dist / dt) * life;
        for (int c = 0; c < 3; ++c) pv[c] = k * D[c];
    }
    for (int c = 0; c < 3; ++c) {
        cf[c] = (v_io[c] - pv[c]) * (p_mass * (1.0 / dt) * P.force_scale);      // :274
        v_io[c] = pv[c];
    }
    return true;
}

// collide_particle :198-231 (penalty contact, collision_type 1).  Returns true inside the band; imp = the impulse p_f dt added to the particle's
// scattered momentum, cf / w as in cloth_collide_mixed.
template <class S>
SMAC_HD bool cloth_collide_particle(const ClothParams& P, const S (*xv)[3], const S (*vv)[3], const S* p_pos, const S* p_v, double dt, int penetrated,
                                    S* imp, S* cf, S* w) {
    S D[3];
    const S dist = cl_sdf_and_normal(p_pos, penetrated, xv[0], xv[1], xv[2], D);
    const S c = dist - 5e-3 * P.scale;                                          // :201-202
    if (!(val(c) < 0.0)) return false;
    S q[3] = {p_pos[0] - D[0] * dist, p_pos[1] - D[1] * dist, p_pos[2] - D[2] * dist};
    cl_barycentric(q, xv[0], xv[1], xv[2], w);                                  // :208
    S in[3];
    for (int k = 0; k < 3; ++k) in[k] = p_v[k] - (w[0] * vv[0][k] + w[1] * vv[1][k] + w[2] * vv[2][k]);   // :210-213
    const S nc = dot3(in, D);
    S t[3] = {in[0] - nc * D[0], in[1] - nc * D[1], in[2] - nc * D[2]};         // :215
    const S tn = sqrt_(dot3(t, t) + 1e-8);                                      // :221
    const S anc = val(nc) < 0.0 ? -nc : nc;                                     // ti.abs
    const double kf = P.friction * 0.001;                                       // :220
    for (int k = 0; k < 3; ++k) {
        const S f1 = -D[k] * c * 140.0;                                         // :217-218
        const S f2 = -t[k] / tn * anc * kf;                                     // :222
        imp[k] = (f1 + f2) * (0.3 * dt);                                        // :224, :231
        cf[k] = -(f1 + f2) * 0.01;                                              // :225
    }
    return true;
}

}  // namespace smac
