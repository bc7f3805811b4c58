// Register-resident rigid-body passes for the "free-flyer + 4 legs x 3 revolute joints" topology
// (Solo12, Go2): every loop has a compile-time trip count and every local array a compile-time
// index, so nothing lives in scratch.  Same quantities and conventions as rbd_device.h /
// oracle/rbd_np.py (which pin them by finite differences); used by the IK-DDP kernels.
//   pass 1  (quad_pass1)   whole robot: CoM, momentum about the origin, centroidal momentum, the
//                          whole-body composite inertia, positions of up to 4 task frames
//   pass 2  (quad_column)  one velocity column: motion subspace S, CoM-Jacobian column, A_g column,
//                          dh_g/dq column (d h_O/dq = S x* h_sub - I_sub (S x V_parent))
#pragma once
#include "rbd_device.h"
#include "lds_batch.h"

#define UNROLL_RBD _Pragma("unroll")

namespace bunmpc {
namespace rbd {

constexpr int kLegs = 4, kLegJoints = 3;

struct BodyAcc { Comp c; double h[6]; };   // composite about the world origin + momentum

RBD_D void rodrigues(const double *a, double q, double *R) {   // unit axis: R = c I + s [a]x + (1 - c) a a^T
    double s, c;
    sincos_fast(q, s, c);
    const double c1 = 1.0 - c;
    R[0] = c + c1 * a[0] * a[0]; R[1] = c1 * a[0] * a[1] - s * a[2]; R[2] = c1 * a[0] * a[2] + s * a[1];
    R[3] = c1 * a[1] * a[0] + s * a[2]; R[4] = c + c1 * a[1] * a[1]; R[5] = c1 * a[1] * a[2] - s * a[0];
    R[6] = c1 * a[2] * a[0] - s * a[1]; R[7] = c1 * a[2] * a[1] + s * a[0]; R[8] = c + c1 * a[2] * a[2];
}

// momentum (l, n_O) only -- no composite inertia: what a cost evaluation needs
RBD_D void body_momentum(const RobotModelDev &m, int b, const double *R, const double *p, const double *V, double &mass,
                         double *h1, double *h) {
    double cw[3], t[3], wl[3], Iw[3], n[3];
    mat3vec(R, m.com[b], cw);
    UNROLL_RBD for (int c = 0; c < 3; ++c) cw[c] += p[c];
    const double mb = m.mass[b];
    cross3(V + 3, cw, t);
    double l[3];
    UNROLL_RBD for (int c = 0; c < 3; ++c) l[c] = mb * (V[c] + t[c]);
    mat3Tvec(R, V + 3, wl);                               // angular velocity in the body frame
    const double *I = m.inertia[b];
    Iw[0] = I[0] * wl[0] + I[1] * wl[1] + I[2] * wl[2];
    Iw[1] = I[1] * wl[0] + I[3] * wl[1] + I[4] * wl[2];
    Iw[2] = I[2] * wl[0] + I[4] * wl[1] + I[5] * wl[2];
    mat3vec(R, Iw, n);
    cross3(cw, l, t);
    mass += mb;
    UNROLL_RBD for (int c = 0; c < 3; ++c) { h1[c] += mb * cw[c]; h[c] += l[c]; h[3 + c] += n[c] + t[c]; }
}

// composite (about the world origin) and momentum of body b placed at (R, p) moving with twist V
template <bool VEL>
RBD_D void body_terms(const RobotModelDev &m, int b, const double *R, const double *p, const double *V, BodyAcc &o) {
    double cw[3], RI[9], Iw[9];
    mat3vec(R, m.com[b], cw);
    UNROLL_RBD for (int c = 0; c < 3; ++c) cw[c] += p[c];
    const double I[9] = {m.inertia[b][0], m.inertia[b][1], m.inertia[b][2], m.inertia[b][1], m.inertia[b][3], m.inertia[b][4],
                         m.inertia[b][2], m.inertia[b][4], m.inertia[b][5]};
    mat3mul(R, I, RI);
    UNROLL_RBD for (int i = 0; i < 3; ++i)
        UNROLL_RBD for (int j = 0; j < 3; ++j)
            Iw[3 * i + j] = RI[3 * i] * R[3 * j] + RI[3 * i + 1] * R[3 * j + 1] + RI[3 * i + 2] * R[3 * j + 2];
    const double mb = m.mass[b], cc = dot3(cw, cw);
    o.c.m = mb;
    UNROLL_RBD for (int c = 0; c < 3; ++c) o.c.h1[c] = mb * cw[c];
    o.c.I[0] = Iw[0] + mb * (cc - cw[0] * cw[0]); o.c.I[1] = Iw[1] - mb * cw[0] * cw[1]; o.c.I[2] = Iw[2] - mb * cw[0] * cw[2];
    o.c.I[3] = Iw[4] + mb * (cc - cw[1] * cw[1]); o.c.I[4] = Iw[5] - mb * cw[1] * cw[2]; o.c.I[5] = Iw[8] + mb * (cc - cw[2] * cw[2]);
    if (VEL) comp_apply(o.c, V, o.h);
}

// one joint of a chain: child placement, motion subspace column, child twist
RBD_D void joint_step(const RobotModelDev &m, int i, double qi, double vi, const double *Rp, const double *pp, const double *Vp,
                      double *R, double *p, double *S, double *V) {
    double Rq[9], t[3];
    rodrigues(m.axis[i], qi, Rq);
    if (m.R_identity[i]) mat3mul(Rp, Rq, R);      // URDF joint origins with rpy = 0 (all of Solo12 / Go2)
    else { double Rl[9]; mat3mul(m.R[i], Rq, Rl); mat3mul(Rp, Rl, R); }
    mat3vec(Rp, m.p[i], t);
    UNROLL_RBD for (int c = 0; c < 3; ++c) p[c] = t[c] + pp[c];
    mat3vec(R, m.axis[i], S + 3);
    cross3(p, S + 3, S);
    UNROLL_RBD for (int c = 0; c < 6; ++c) V[c] = Vp[c] + S[c] * vi;
}

// ---- body constants in registers: one batched LDS read per body instead of ~16 scattered ones (the kernels keep the
// model in LDS; hipcc glues each scattered ds_read to its first use, one exposed latency apiece)
struct BodyRec { double p[3], axis[3], mass, com[3], I[6]; };

RBD_D void load_body(const RobotModelDev &m, int b, BodyRec &o) {   // m must live in LDS
    double2_t t[8];
    lds_read_b128x8(lds_offset(m.rec[b]), t);
    o.p[0] = t[0].x; o.p[1] = t[0].y; o.p[2] = t[1].x; o.axis[0] = t[1].y; o.axis[1] = t[2].x; o.axis[2] = t[2].y;
    o.mass = t[3].x; o.com[0] = t[3].y; o.com[1] = t[4].x; o.com[2] = t[4].y;
    o.I[0] = t[5].x; o.I[1] = t[5].y; o.I[2] = t[6].x; o.I[3] = t[6].y; o.I[4] = t[7].x; o.I[5] = t[7].y;
}

RBD_D void body_momentum_r(const BodyRec &br, const double *R, const double *p, const double *V, double &mass, double *h1, double *h) {
    double cw[3], t[3], wl[3], Iw[3], n[3];
    mat3vec(R, br.com, cw);
    UNROLL_RBD for (int c = 0; c < 3; ++c) cw[c] += p[c];
    const double mb = br.mass;
    cross3(V + 3, cw, t);
    double l[3];
    UNROLL_RBD for (int c = 0; c < 3; ++c) l[c] = mb * (V[c] + t[c]);
    mat3Tvec(R, V + 3, wl);
    const double *I = br.I;
    Iw[0] = I[0] * wl[0] + I[1] * wl[1] + I[2] * wl[2];
    Iw[1] = I[1] * wl[0] + I[3] * wl[1] + I[4] * wl[2];
    Iw[2] = I[2] * wl[0] + I[4] * wl[1] + I[5] * wl[2];
    mat3vec(R, Iw, n);
    cross3(cw, l, t);
    double mc[3], nt[3];
    UNROLL_RBD for (int c = 0; c < 3; ++c) { mc[c] = mb * cw[c]; nt[c] = n[c] + t[c]; }
    {   // the body's terms join the running sums by plain additions: fused into the products above they would round differently
        // from the sixteen-lane walk (quad_part16), which adds the same terms across lanes
#pragma clang fp contract(off)
        mass += mb;
        UNROLL_RBD for (int c = 0; c < 3; ++c) { h1[c] += mc[c]; h[c] += l[c]; h[3 + c] += nt[c]; }
    }
}

template <bool VEL>
RBD_D void body_terms_r(const BodyRec &br, const double *R, const double *p, const double *V, BodyAcc &o) {
    double cw[3], RI[9], Iw[9];
    mat3vec(R, br.com, cw);
    UNROLL_RBD for (int c = 0; c < 3; ++c) cw[c] += p[c];
    const double I[9] = {br.I[0], br.I[1], br.I[2], br.I[1], br.I[3], br.I[4], br.I[2], br.I[4], br.I[5]};
    mat3mul(R, I, RI);
    UNROLL_RBD for (int i = 0; i < 3; ++i)
        UNROLL_RBD for (int j = 0; j < 3; ++j)
            Iw[3 * i + j] = RI[3 * i] * R[3 * j] + RI[3 * i + 1] * R[3 * j + 1] + RI[3 * i + 2] * R[3 * j + 2];
    const double mb = br.mass, cc = dot3(cw, cw);
    o.c.m = mb;
    UNROLL_RBD for (int c = 0; c < 3; ++c) o.c.h1[c] = mb * cw[c];
    o.c.I[0] = Iw[0] + mb * (cc - cw[0] * cw[0]); o.c.I[1] = Iw[1] - mb * cw[0] * cw[1]; o.c.I[2] = Iw[2] - mb * cw[0] * cw[2];
    o.c.I[3] = Iw[4] + mb * (cc - cw[1] * cw[1]); o.c.I[4] = Iw[5] - mb * cw[1] * cw[2]; o.c.I[5] = Iw[8] + mb * (cc - cw[2] * cw[2]);
    if (VEL) comp_apply(o.c, V, o.h);
}

RBD_D void joint_step_r(const RobotModelDev &m, const BodyRec &br, int i, double qi, double vi, const double *Rp, const double *pp,
                        const double *Vp, double *R, double *p, double *S, double *V) {
    double Rq[9], t[3];
    rodrigues(br.axis, qi, Rq);
    if (m.R_identity[i]) mat3mul(Rp, Rq, R);
    else { double Rl[9]; mat3mul(m.R[i], Rq, Rl); mat3mul(Rp, Rl, R); }
    mat3vec(Rp, br.p, t);
    UNROLL_RBD for (int c = 0; c < 3; ++c) p[c] = t[c] + pp[c];
    mat3vec(R, br.axis, S + 3);
    cross3(p, S + 3, S);
    UNROLL_RBD for (int c = 0; c < 6; ++c) V[c] = Vp[c] + S[c] * vi;
}

struct Pass1 {
    double Rb[9], pb[3], Vb[6];
    Comp call;                 // whole-robot composite about the origin
    double hO[6], com[3], hg[6], M;
    double fx[kFrameSlots][3]; // task-frame positions
};

// fid[s] < 0: slot unused.  COMPOSITE = false skips the whole-body composite inertia (cost-only passes).
template <bool COMPOSITE>
RBD_D void quad_pass1(const RobotModelDev &m, const double *x, const int *fid, Pass1 &o) {
    quat_to_R(x + 3, o.Rb);
    o.pb[0] = x[0]; o.pb[1] = x[1]; o.pb[2] = x[2];
    const double *v = x + kNQ;
    {   // base twist about the world origin: v_O = R v_lin + p x (R w)
        double wl[3], vl[3], t[3];
        mat3vec(o.Rb, v + 3, wl); mat3vec(o.Rb, v, vl); cross3(o.pb, wl, t);
        UNROLL_RBD for (int c = 0; c < 3; ++c) { o.Vb[c] = vl[c] + t[c]; o.Vb[3 + c] = wl[c]; }
    }
    int fbody[kFrameSlots];
    UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) {
        fbody[s] = fid[s] >= 0 ? m.frame_body[fid[s]] : -1;
        o.fx[s][0] = o.fx[s][1] = o.fx[s][2] = 0.0;
    }
    auto frames_on = [&](int b, const double *R, const double *p) {
        UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s)
            if (fbody[s] == b) {
                double t[3];
                mat3vec(R, m.frame_p[fid[s]], t);
                UNROLL_RBD for (int c = 0; c < 3; ++c) o.fx[s][c] = t[c] + p[c];
            }
    };
    BodyAcc ba;
    double h1[3] = {0, 0, 0};
    o.M = 0.0;
    UNROLL_RBD for (int c = 0; c < 6; ++c) o.hO[c] = 0.0;
    if (COMPOSITE) {
        body_terms<true>(m, 0, o.Rb, o.pb, o.Vb, ba);
        o.call = ba.c;
        UNROLL_RBD for (int c = 0; c < 6; ++c) o.hO[c] = ba.h[c];
    } else body_momentum(m, 0, o.Rb, o.pb, o.Vb, o.M, h1, o.hO);
    frames_on(0, o.Rb, o.pb);
    UNROLL_RBD for (int L = 0; L < kLegs; ++L) {
        double Rp[9], pp[3], Vp[6];
        UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = o.Rb[c];
        UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = o.pb[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = o.Vb[c];
        UNROLL_RBD for (int j = 0; j < kLegJoints; ++j) {
            const int i = kLegJoints * L + j;
            double R[9], p[3], S[6], V[6];
            joint_step(m, i, x[7 + i], v[6 + i], Rp, pp, Vp, R, p, S, V);
            if (COMPOSITE) {
                body_terms<true>(m, i + 1, R, p, V, ba);
                comp_add(o.call, ba.c);
                UNROLL_RBD for (int c = 0; c < 6; ++c) o.hO[c] += ba.h[c];
            } else body_momentum(m, i + 1, R, p, V, o.M, h1, o.hO);
            frames_on(i + 1, R, p);
            UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = R[c];
            UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = p[c];
            UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = V[c];
        }
    }
    if (COMPOSITE) { o.M = o.call.m; UNROLL_RBD for (int c = 0; c < 3; ++c) h1[c] = o.call.h1[c]; }
    UNROLL_RBD for (int c = 0; c < 3; ++c) o.com[c] = h1[c] / o.M;
    double t[3];
    cross3(o.com, o.hO, t);
    UNROLL_RBD for (int c = 0; c < 3; ++c) { o.hg[c] = o.hO[c]; o.hg[3 + c] = o.hO[3 + c] - t[c]; }
}

// Partial sums of one part of the robot (part 0..3 = leg, part 4 = base body): mass, first moment,
// momentum about the origin, and the position of task frames carried by that part.  Lets a wave
// spread one node evaluation over five lanes (forward rollout) and add the parts in LDS.
struct PartSum { double mass, h1[3], hO[6], fx[kFrameSlots][3]; int fhit[kFrameSlots]; };

// x: the state with compile-time indices only (registers); qj / vj: the three joint angles / rates of leg `part`
RBD_D void quad_part(const RobotModelDev &m, const double *x, const double *qj, const double *vj, const int *fid, int part, PartSum &o) {
    double Rb[9], pb[3], Vb[6];
    quat_to_R(x + 3, Rb);
    pb[0] = x[0]; pb[1] = x[1]; pb[2] = x[2];
    const double *v = x + kNQ;
    {
        double wl[3], vl[3], t[3];
        mat3vec(Rb, v + 3, wl); mat3vec(Rb, v, vl); cross3(pb, wl, t);
        UNROLL_RBD for (int c = 0; c < 3; ++c) { Vb[c] = vl[c] + t[c]; Vb[3 + c] = wl[c]; }
    }
    o.mass = 0.0;
    UNROLL_RBD for (int c = 0; c < 3; ++c) o.h1[c] = 0.0;
    UNROLL_RBD for (int c = 0; c < 6; ++c) o.hO[c] = 0.0;
    int fbody[kFrameSlots];
    UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) {
        fbody[s] = fid[s] >= 0 ? m.frame_body[fid[s]] : -1;
        o.fhit[s] = 0;
        o.fx[s][0] = o.fx[s][1] = o.fx[s][2] = 0.0;
    }
    auto frames_on = [&](int b, const double *R, const double *p) {
        UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s)
            if (fbody[s] == b) {
                double t[3];
                mat3vec(R, m.frame_p[fid[s]], t);
                UNROLL_RBD for (int c = 0; c < 3; ++c) o.fx[s][c] = t[c] + p[c];
                o.fhit[s] = 1;
            }
    };
    BodyRec br;
    if (part == kLegs) {
        load_body(m, 0, br);
        body_momentum_r(br, Rb, pb, Vb, o.mass, o.h1, o.hO);
        frames_on(0, Rb, pb);
        return;
    }
    double Rp[9], pp[3], Vp[6];
    UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = Rb[c];
    UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = pb[c];
    UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = Vb[c];
    UNROLL_RBD for (int j = 0; j < kLegJoints; ++j) {
        const int i = kLegJoints * part + j;      // runtime leg: model / state reads are indexed, locals are not
        double R[9], p[3], S[6], V[6];
        load_body(m, i + 1, br);
        joint_step_r(m, br, i, qj[j], vj[j], Rp, pp, Vp, R, p, S, V);
        body_momentum_r(br, R, p, V, o.mass, o.h1, o.hO);
        frames_on(i + 1, R, p);
        UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = R[c];
        UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = p[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = V[c];
    }
}

// ---- the same part sums with SIXTEEN lanes of a sub-group: lane l = 4 leg + j, j < 3 the joints of the leg, lane 3 the base
// body (the spare lanes of the other quads idle).  quad_part walks a leg joint after joint on one lane; where the walk has a
// wave of its own beside the rollout chain (the few-problems mappings of the forward pass) that scalar chain sets the pace of
// a node (~15.8K cycles against ~14.6K of the chain).  Here every lane prepares its own joint (sincos, local rotation, body
// constants) at once; only the composition of the world placement runs down the leg, each step computed by all lanes and
// handed on inside the quad by DPP (quad_perm broadcast); then every lane takes the momentum of its own body and the quad
// butterfly adds the three bodies of a leg -- (b0 + b1) + b2, the order quad_part accumulates in, so the sums carry the same
// bits.  ~11.2K cycles per node there; in the many-problems mapping (everything on one wave, issue bound) the extra
// instructions of the cooperative version cost more than they save, and quad_part stays.
template <int CTRL>
RBD_D double quad_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
RBD_D double quad_sum(double v) {      // (v0 + v1) + (v2 + v3) on every lane of the quad
    v += quad_dpp<0xB1>(v);            // quad_perm:[1,0,3,2]
    v += quad_dpp<0x4E>(v);            // quad_perm:[2,3,0,1]
    return v;
}
// xs: the state in LDS (joint values are read by run-time index); o of lane 4 leg (j = 0): the leg's sums; o of lane 3: the base body's
RBD_D void quad_part16(const RobotModelDev &m, const double *xs, const int *fid, int l, PartSum &o) {
    const int leg = l >> 2, jj = l & 3;
    const bool is_joint = jj < kLegJoints, is_base = l == 3;
    const int i = kLegJoints * leg + (is_joint ? jj : 0);
    double Rb[9], pb[3], Vb[6];
    {
        double xv[40];
        double2_t t[20];
        lds_read_b128x20(lds_offset(xs), t);
        UNROLL_RBD for (int k = 0; k < 20; ++k) { xv[2 * k] = t[k].x; xv[2 * k + 1] = t[k].y; }
        quat_to_R(xv + 3, Rb);
        pb[0] = xv[0]; pb[1] = xv[1]; pb[2] = xv[2];
        const double *v = xv + kNQ;
        double wl[3], vl[3], tt[3];
        mat3vec(Rb, v + 3, wl); mat3vec(Rb, v, vl); cross3(pb, wl, tt);
        UNROLL_RBD for (int c = 0; c < 3; ++c) { Vb[c] = vl[c] + tt[c]; Vb[3 + c] = wl[c]; }
    }
    const double qi = is_joint ? xs[7 + i] : 0.0, vi = is_joint ? xs[kNQ + 6 + i] : 0.0;
    BodyRec br;
    load_body(m, is_joint ? i + 1 : 0, br);
    double Rl[9];
    rodrigues(br.axis, qi, Rl);                      // the base lane (axis of body 0 = 0, q = 0) gets the identity
    if (__any(is_joint && !m.R_identity[i])) {       // joint placements with a rotation: none on Solo12 / Go2
        double t9[9];
        UNROLL_RBD for (int c = 0; c < 9; ++c) t9[c] = Rl[c];
        if (is_joint && !m.R_identity[i]) mat3mul(m.R[i], t9, Rl);
    }
    double cur[9], pp[3], Vp[6], R[9], p[3], V[6];
    UNROLL_RBD for (int c = 0; c < 9; ++c) { cur[c] = Rb[c]; R[c] = Rb[c]; }
    UNROLL_RBD for (int c = 0; c < 3; ++c) { pp[c] = pb[c]; p[c] = pb[c]; }
    UNROLL_RBD for (int c = 0; c < 6; ++c) { Vp[c] = Vb[c]; V[c] = Vb[c]; }
    UNROLL_RBD for (int k = 0; k < kLegJoints; ++k) {
        double cR[9], cp[3], S[6], t3[3];
        mat3mul(cur, Rl, cR);
        mat3vec(cur, br.p, t3);
        UNROLL_RBD for (int c = 0; c < 3; ++c) cp[c] = t3[c] + pp[c];
        mat3vec(cR, br.axis, S + 3);
        cross3(cp, S + 3, S);
        const bool mine = jj == k;
        UNROLL_RBD for (int c = 0; c < 9; ++c) R[c] = mine ? cR[c] : R[c];
        UNROLL_RBD for (int c = 0; c < 3; ++c) p[c] = mine ? cp[c] : p[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) V[c] = mine ? Vp[c] + S[c] * vi : V[c];
        if (k == 0) {
            UNROLL_RBD for (int c = 0; c < 9; ++c) cur[c] = quad_dpp<0x00>(R[c]);
            UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = quad_dpp<0x00>(p[c]);
            UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = quad_dpp<0x00>(V[c]);
        } else if (k == 1) {
            UNROLL_RBD for (int c = 0; c < 9; ++c) cur[c] = quad_dpp<0x55>(R[c]);
            UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = quad_dpp<0x55>(p[c]);
            UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = quad_dpp<0x55>(V[c]);
        }
    }
    double mass = 0.0, h1[3] = {0, 0, 0}, hO[6] = {0, 0, 0, 0, 0, 0};
    body_momentum_r(br, R, p, V, mass, h1, hO);
    const int my_body = is_joint ? i + 1 : (is_base ? 0 : -1);
    double fx[kFrameSlots][3];
    int hit[kFrameSlots];
    UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) {
        const int fb = fid[s] >= 0 ? m.frame_body[fid[s]] : -2;
        hit[s] = fb == my_body;
        double t3[3] = {0, 0, 0};
        if (__any(hit[s])) {
            const double *fp = m.frame_p[fid[s] >= 0 ? fid[s] : 0];
            mat3vec(R, fp, t3);
        }
        UNROLL_RBD for (int c = 0; c < 3; ++c) fx[s][c] = hit[s] ? t3[c] + p[c] : 0.0;
    }
    if (is_base) {       // the base body's own terms, before the leg sums mix the quad
        o.mass = mass;
        UNROLL_RBD for (int c = 0; c < 3; ++c) o.h1[c] = h1[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) o.hO[c] = hO[c];
        UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) { o.fhit[s] = hit[s]; UNROLL_RBD for (int c = 0; c < 3; ++c) o.fx[s][c] = fx[s][c]; }
    }
    const double keep = is_joint ? 1.0 : 0.0;
    const double ms = quad_sum(keep * mass);
    double h1s[3], hOs[6];
    UNROLL_RBD for (int c = 0; c < 3; ++c) h1s[c] = quad_sum(keep * h1[c]);
    UNROLL_RBD for (int c = 0; c < 6; ++c) hOs[c] = quad_sum(keep * hO[c]);
    if (!is_base) {
        o.mass = ms;
        UNROLL_RBD for (int c = 0; c < 3; ++c) o.h1[c] = h1s[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) o.hO[c] = hOs[c];
    }
    UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) {
        const double hs = quad_sum(is_joint && hit[s] ? 1.0 : 0.0);
        double f3[3];
        UNROLL_RBD for (int c = 0; c < 3; ++c) f3[c] = quad_sum(is_joint ? fx[s][c] : 0.0);
        if (!is_base) { o.fhit[s] = hs != 0.0; UNROLL_RBD for (int c = 0; c < 3; ++c) o.fx[s][c] = f3[c]; }
    }
}

struct Column { double S[6], jc[3], ag[6], dh[6]; };

// velocity column `col` (0..17) at state x; p1 = pass 1 of the same state
RBD_D void quad_column(const RobotModelDev &m, const double *x, const Pass1 &p1, int col, Column &o) {
    Comp cs; double hs[6], Vpar[6];
    if (col < 6) {
        const int a = col % 3;
        const double e[3] = {p1.Rb[a], p1.Rb[3 + a], p1.Rb[6 + a]};
        if (col < 3) { UNROLL_RBD for (int c = 0; c < 3; ++c) { o.S[c] = e[c]; o.S[3 + c] = 0.0; } }
        else { cross3(p1.pb, e, o.S); UNROLL_RBD for (int c = 0; c < 3; ++c) o.S[3 + c] = e[c]; }
        cs = p1.call;
        UNROLL_RBD for (int c = 0; c < 6; ++c) { hs[c] = p1.hO[c]; Vpar[c] = 0.0; }
    } else {
        const int L = (col - 6) / kLegJoints, jsel = (col - 6) % kLegJoints;
        const double *v = x + kNQ;
        double Rp[9], pp[3], Vp[6];
        UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = p1.Rb[c];
        UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = p1.pb[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = p1.Vb[c];
        comp_zero(cs);
        UNROLL_RBD for (int c = 0; c < 6; ++c) { hs[c] = 0.0; Vpar[c] = 0.0; o.S[c] = 0.0; }
        UNROLL_RBD for (int j = 0; j < kLegJoints; ++j) {
            const int i = kLegJoints * L + j;
            double R[9], p[3], S[6], V[6];
            joint_step(m, i, x[7 + i], v[6 + i], Rp, pp, Vp, R, p, S, V);
            BodyAcc ba;
            body_terms<true>(m, i + 1, R, p, V, ba);
            if (j == jsel) { UNROLL_RBD for (int c = 0; c < 6; ++c) { o.S[c] = S[c]; Vpar[c] = Vp[c]; } }
            if (j >= jsel) { comp_add(cs, ba.c); UNROLL_RBD for (int c = 0; c < 6; ++c) hs[c] += ba.h[c]; }   // subtree = joints jsel..2
            UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = R[c];
            UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = p[c];
            UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = V[c];
        }
    }
    double h[6], t3[3], a3[3], b3[3];
    comp_apply(cs, o.S, h);
    UNROLL_RBD for (int c = 0; c < 3; ++c) o.jc[c] = h[c] / p1.M;
    cross3(p1.com, h, t3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) { o.ag[c] = h[c]; o.ag[3 + c] = h[3 + c] - t3[c]; }
    double cf[6], sxv[6], ih[6];
    cross3(o.S + 3, hs, cf);
    cross3(o.S + 3, hs + 3, a3); cross3(o.S, hs, b3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) cf[3 + c] = a3[c] + b3[c];
    cross3(o.S + 3, Vpar, a3); cross3(o.S, Vpar + 3, b3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) sxv[c] = a3[c] + b3[c];
    cross3(o.S + 3, Vpar + 3, sxv + 3);
    comp_apply(cs, sxv, ih);
    double dO[6];
    UNROLL_RBD for (int c = 0; c < 6; ++c) dO[c] = cf[c] - ih[c];
    cross3(o.jc, p1.hO, a3); cross3(p1.com, dO, b3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) { o.dh[c] = dO[c]; o.dh[3 + c] = dO[3 + c] - a3[c] - b3[c]; }
}

// ---- one chain walk per lane (calcDiff): lanes 0..5 own the base columns, lane 6 + 3 L + jsel joint jsel of leg L.
// A lane walks only its own part of the robot and leaves (a) that part's composite / momentum about the origin
// (to be added over the five parts: base, legs), (b) what its own column needs: motion subspace S, the composite and
// momentum of the subtree the column moves, the parent twist.  quad_col_finish turns (b) + the totals into the column.
struct PartWalk {
    Comp part; double hpart[6];            // this lane's part of the robot (base body or whole leg)
    Comp cs; double hs[6], S[6], Vpar[6];  // column data; for base columns cs / hs are the robot totals (set by the caller)
    double fx[kFrameSlots][3];             // positions of the task frames carried by this part
    int fhit[kFrameSlots];                 // ... and which of them are
};

RBD_D void quad_part_walk(const RobotModelDev &m, const double *x, const int *fid, int col, double *Rb, double *pb, double *Vb, PartWalk &o) {
    quat_to_R(x + 3, Rb);
    pb[0] = x[0]; pb[1] = x[1]; pb[2] = x[2];
    const double *v = x + kNQ;
    {
        double wl[3], vl[3], t[3];
        mat3vec(Rb, v + 3, wl); mat3vec(Rb, v, vl); cross3(pb, wl, t);
        UNROLL_RBD for (int c = 0; c < 3; ++c) { Vb[c] = vl[c] + t[c]; Vb[3 + c] = wl[c]; }
    }
    int fbody[kFrameSlots];
    UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s) {
        fbody[s] = fid[s] >= 0 ? m.frame_body[fid[s]] : -1;
        o.fx[s][0] = o.fx[s][1] = o.fx[s][2] = 0.0;
        o.fhit[s] = 0;
    }
    auto frames_on = [&](int b, const double *R, const double *p) {
        UNROLL_RBD for (int s = 0; s < kFrameSlots; ++s)
            if (fbody[s] == b) {
                double t[3];
                mat3vec(R, m.frame_p[fid[s]], t);
                UNROLL_RBD for (int c = 0; c < 3; ++c) o.fx[s][c] = t[c] + p[c];
                o.fhit[s] = 1;
            }
    };
    BodyAcc ba;
    BodyRec br;
    if (col < 6) {
        load_body(m, 0, br);
        body_terms_r<true>(br, Rb, pb, Vb, ba);
        o.part = ba.c;
        UNROLL_RBD for (int c = 0; c < 6; ++c) { o.hpart[c] = ba.h[c]; o.Vpar[c] = 0.0; }
        frames_on(0, Rb, pb);
        const int a = col % 3;
        const double e[3] = {Rb[a], Rb[3 + a], Rb[6 + a]};
        if (col < 3) { UNROLL_RBD for (int c = 0; c < 3; ++c) { o.S[c] = e[c]; o.S[3 + c] = 0.0; } }
        else { cross3(pb, e, o.S); UNROLL_RBD for (int c = 0; c < 3; ++c) o.S[3 + c] = e[c]; }
        return;
    }
    const int L = (col - 6) / kLegJoints, jsel = (col - 6) % kLegJoints;
    double Rp[9], pp[3], Vp[6];
    UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = Rb[c];
    UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = pb[c];
    UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = Vb[c];
    comp_zero(o.part); comp_zero(o.cs);
    UNROLL_RBD for (int c = 0; c < 6; ++c) { o.hpart[c] = 0.0; o.hs[c] = 0.0; o.Vpar[c] = 0.0; o.S[c] = 0.0; }
    UNROLL_RBD for (int j = 0; j < kLegJoints; ++j) {
        const int i = kLegJoints * L + j;
        double R[9], p[3], S[6], V[6];
        load_body(m, i + 1, br);
        joint_step_r(m, br, i, x[7 + i], v[6 + i], Rp, pp, Vp, R, p, S, V);
        body_terms_r<true>(br, R, p, V, ba);
        comp_add(o.part, ba.c);
        UNROLL_RBD for (int c = 0; c < 6; ++c) o.hpart[c] += ba.h[c];
        if (j == jsel) { UNROLL_RBD for (int c = 0; c < 6; ++c) { o.S[c] = S[c]; o.Vpar[c] = Vp[c]; } }
        if (j >= jsel) { comp_add(o.cs, ba.c); UNROLL_RBD for (int c = 0; c < 6; ++c) o.hs[c] += ba.h[c]; }   // subtree = joints jsel..2
        frames_on(i + 1, R, p);
        UNROLL_RBD for (int c = 0; c < 9; ++c) Rp[c] = R[c];
        UNROLL_RBD for (int c = 0; c < 3; ++c) pp[c] = p[c];
        UNROLL_RBD for (int c = 0; c < 6; ++c) Vp[c] = V[c];
    }
}

// column from the walk + robot totals (M, com, hO): CoM-Jacobian column, A_g column, dh_g/dq column
RBD_D void quad_col_finish(const PartWalk &w, double M, const double *com, const double *hO, Column &o) {
    double h[6], t3[3], a3[3], b3[3];
    UNROLL_RBD for (int c = 0; c < 6; ++c) o.S[c] = w.S[c];
    comp_apply(w.cs, o.S, h);
    const double iM = 1.0 / M;
    UNROLL_RBD for (int c = 0; c < 3; ++c) o.jc[c] = h[c] * iM;
    cross3(com, h, t3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) { o.ag[c] = h[c]; o.ag[3 + c] = h[3 + c] - t3[c]; }
    double cf[6], sxv[6], ih[6];
    cross3(o.S + 3, w.hs, cf);
    cross3(o.S + 3, w.hs + 3, a3); cross3(o.S, w.hs, b3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) cf[3 + c] = a3[c] + b3[c];
    cross3(o.S + 3, w.Vpar, a3); cross3(o.S, w.Vpar + 3, b3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) sxv[c] = a3[c] + b3[c];
    cross3(o.S + 3, w.Vpar + 3, sxv + 3);
    comp_apply(w.cs, sxv, ih);
    double dO[6];
    UNROLL_RBD for (int c = 0; c < 6; ++c) dO[c] = cf[c] - ih[c];
    cross3(o.jc, hO, a3); cross3(com, dO, b3);
    UNROLL_RBD for (int c = 0; c < 3; ++c) { o.dh[c] = dO[c]; o.dh[3 + c] = dO[3 + c] - a3[c] - b3[c]; }
}

// does column col move the body that carries frame f?  (legs numbered 3L..3L+2, body = joint + 1)
RBD_D bool quad_supports(const RobotModelDev &m, int f, int col) {
    if (col < 6) return true;
    const int b = m.frame_body[f];
    if (b == 0) return false;
    const int jf = b - 1, jc = col - 6;
    return (jf / kLegJoints == jc / kLegJoints) && (jc <= jf);
}

}  // namespace rbd
}  // namespace bunmpc
