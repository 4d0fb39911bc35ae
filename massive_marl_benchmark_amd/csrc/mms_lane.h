// mms_lane.h -- per-lane math of the step kernels (gfx950), written so that the same source also
// compiles for the host: tests/emu builds it with g++ to check the lane decomposition against the
// oracle on the CPU before any GPU time is spent.  No HIP builtins in this file.
//
// Decomposition: one workgroup per environment.  Lane (ant a, leg l) = 4a + l owns
// one leg chain (leg + foot bodies, 2 DOF, 3 contact spheres) and a replica of its ant's torso state;
// the 4 lanes of a quad combine their articulated inertias with two DPP quad-permutes; 8 further
// lanes own the box corners.  Math follows oracle/mms_oracle.c (the model definition) term by term;
// reference call sites are cited there.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/mms.h"

#if defined(__HIPCC__)
#define MMS_HD __host__ __device__ __forceinline__
#else
#define MMS_HD inline
#endif
// (rare paths -- an ant in reach of the box: block frequencies steer the register allocator's spill weights and the block layout)
#define MMS_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace mms {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kContactMargin = 0.1f;
constexpr float kMaxAngVel = 64.f;

// ---------------------------------------------------------------------------------------------
// small vectors
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
MMS_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
MMS_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MMS_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MMS_HD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
MMS_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MMS_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
MMS_HD float clampf(float x, float lo, float hi) { return fmaxf(fminf(x, hi), lo); }
// activation weight of contacts / joint limits: clamp(x / r, 0, 1) (continuous contact law, see the oracle)
MMS_HD float ramp01(float d, float r) { return fminf(fmaxf(d / r, 0.f), 1.f); }

struct M3 { V3 c0, c1, c2; };   // columns
MMS_HD V3 mul(const M3& R, V3 v) { return v.x * R.c0 + v.y * R.c1 + v.z * R.c2; }
MMS_HD V3 mulT(const M3& R, V3 v) { return V3{dot(R.c0, v), dot(R.c1, v), dot(R.c2, v)}; }
MMS_HD M3 mul(const M3& A, const M3& B) { return M3{mul(A, B.c0), mul(A, B.c1), mul(A, B.c2)}; }
MMS_HD M3 quat_to_mat(float x, float y, float z, float w) {
    M3 R;
    R.c0 = V3{1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y)};
    R.c1 = V3{2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x)};
    R.c2 = V3{2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y)};
    return R;
}
MMS_HD M3 axis_angle_to_mat(V3 a, float ang) {
    float c = cosf(ang), s = sinf(ang), t = 1.f - c;
    M3 R;
    R.c0 = V3{c + t * a.x * a.x, t * a.x * a.y + s * a.z, t * a.x * a.z - s * a.y};
    R.c1 = V3{t * a.x * a.y - s * a.z, c + t * a.y * a.y, t * a.y * a.z + s * a.x};
    R.c2 = V3{t * a.x * a.z + s * a.y, t * a.y * a.z - s * a.x, c + t * a.z * a.z};
    return R;
}

// spatial 6-vector: angular part a, linear part l
struct S6 { V3 a, l; };
MMS_HD S6 operator+(S6 p, S6 q) { return S6{p.a + q.a, p.l + q.l}; }
MMS_HD S6 operator*(float s, S6 p) { return S6{s * p.a, s * p.l}; }
MMS_HD float dot(S6 p, S6 q) { return dot(p.a, q.a) + dot(p.l, q.l); }
MMS_HD float get(const S6& p, int i) { return i == 0 ? p.a.x : i == 1 ? p.a.y : i == 2 ? p.a.z : i == 3 ? p.l.x : i == 4 ? p.l.y : p.l.z; }
MMS_HD S6 cross_motion(S6 a, S6 b) { return S6{cross(a.a, b.a), cross(a.a, b.l) + cross(a.l, b.a)}; }

// symmetric 6x6, upper triangle row-major (21 values); every index below is a compile-time constant
// once the loops are unrolled, so the matrix lives in registers.
constexpr int sidx(int i, int j) { return i <= j ? i * 6 - i * (i - 1) / 2 + (j - i) : j * 6 - j * (j - 1) / 2 + (i - j); }
struct Sym6 { float m[21]; };
MMS_HD void sym_zero(Sym6& A) {
#pragma unroll
    for (int k = 0; k < 21; k++) A.m[k] = 0.f;
}
MMS_HD void sym_add(Sym6& A, const Sym6& B) {
#pragma unroll
    for (int k = 0; k < 21; k++) A.m[k] += B.m[k];
}
MMS_HD S6 sym_mul(const Sym6& A, S6 v) {
    float in[6] = {v.a.x, v.a.y, v.a.z, v.l.x, v.l.y, v.l.z}, o[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 6; j++) s += A.m[sidx(i, j)] * in[j];
        o[i] = s;
    }
    return S6{V3{o[0], o[1], o[2]}, V3{o[3], o[4], o[5]}};
}
// A += s * w w^T
MMS_HD void sym_rank1(Sym6& A, float s, S6 w) {
    float in[6] = {w.a.x, w.a.y, w.a.z, w.l.x, w.l.y, w.l.z};
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) A.m[sidx(i, j)] += s * in[i] * in[j];
}

// spatial inertia about the frame origin of an axisymmetric body: mass m, COM c, unit axis u,
// axial / transverse inertia ia / it about the COM
MMS_HD void spatial_inertia_axisym(float m, V3 c, V3 u, float ia, float it, Sym6& I) {
    float cc = dot(c, c), d = ia - it;
    float cv[3] = {c.x, c.y, c.z}, uv[3] = {u.x, u.y, u.z};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) {
            float ic = (i == j ? it : 0.f) + d * uv[i] * uv[j];
            I.m[sidx(i, j)] = ic + m * ((i == j ? cc : 0.f) - cv[i] * cv[j]);
            I.m[sidx(3 + i, 3 + j)] = (i == j) ? m : 0.f;
        }
    // top-right block m [c]x
    I.m[sidx(0, 3)] = 0.f;        I.m[sidx(0, 4)] = -m * c.z;  I.m[sidx(0, 5)] = m * c.y;
    I.m[sidx(1, 3)] = m * c.z;    I.m[sidx(1, 4)] = 0.f;       I.m[sidx(1, 5)] = -m * c.x;
    I.m[sidx(2, 3)] = -m * c.y;   I.m[sidx(2, 4)] = m * c.x;   I.m[sidx(2, 5)] = 0.f;
}
// general symmetric 3x3 rotational inertia (box, helicopter): Iw = R diag(d) R^T
MMS_HD void spatial_inertia_diag(float m, V3 c, const M3& R, V3 d, Sym6& I) {
    float cc = dot(c, c);
    float cv[3] = {c.x, c.y, c.z};
    float r0[3] = {R.c0.x, R.c1.x, R.c2.x}, r1[3] = {R.c0.y, R.c1.y, R.c2.y}, r2[3] = {R.c0.z, R.c1.z, R.c2.z};
    const float* rows[3] = {r0, r1, r2};
    float dv[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) {
            float ic = rows[i][0] * dv[0] * rows[j][0] + rows[i][1] * dv[1] * rows[j][1] + rows[i][2] * dv[2] * rows[j][2];
            I.m[sidx(i, j)] = ic + m * ((i == j ? cc : 0.f) - cv[i] * cv[j]);
            I.m[sidx(3 + i, 3 + j)] = (i == j) ? m : 0.f;
        }
    I.m[sidx(0, 3)] = 0.f;        I.m[sidx(0, 4)] = -m * c.z;  I.m[sidx(0, 5)] = m * c.y;
    I.m[sidx(1, 3)] = m * c.z;    I.m[sidx(1, 4)] = 0.f;       I.m[sidx(1, 5)] = -m * c.x;
    I.m[sidx(2, 3)] = -m * c.y;   I.m[sidx(2, 4)] = m * c.x;   I.m[sidx(2, 5)] = 0.f;
}
// bias force p = v x* (I v) - (c x m g, m g),  g = (0, 0, -grav)
MMS_HD S6 bias_force(const Sym6& I, S6 v, float m, V3 c, float grav) {
    S6 h = sym_mul(I, v);
    V3 fg = V3{0.f, 0.f, -m * grav};
    V3 ng = cross(c, fg);
    S6 p;
    p.a = cross(v.a, h.a) + cross(v.l, h.l) - ng;
    p.l = cross(v.a, h.l) - fg;
    return p;
}
// the same for an axisymmetric body, from its parameters instead of the assembled 6x6 (momentum: f = m (vO + w x c),
// n = Ic w + c x f with Ic w = it w + (ia - it)(u.w) u): about half the instructions of the dense product
MMS_HD S6 bias_force_axisym(S6 v, float m, V3 c, V3 u, float ia, float it, float grav) {
    V3 f = m * (v.l + cross(v.a, c));
    V3 n = it * v.a + ((ia - it) * dot(u, v.a)) * u + cross(c, f);
    V3 fg = V3{0.f, 0.f, -m * grav};
    S6 p;
    p.a = cross(v.a, n) + cross(v.l, f) - cross(c, fg);
    p.l = cross(v.a, f) - fg;
    return p;
}
// LDL^T of a symmetric positive definite 6x6 system, split into the factorisation (matrix only) and the two substitutions
// (right-hand side): the box's matrix is known before its right-hand side is (the ants' reaction arrives through a barrier), so
// the step kernel factors before the barrier and only substitutes behind it.  solve6 = both, the same arithmetic in the same order.
struct Ldl6 { float L[15]; float D[6]; };     // L[i][j], i > j, at lidx(i, j); D = the pivots
constexpr int lidx(int i, int j) { return i * (i - 1) / 2 + j; }
MMS_HD Ldl6 factor6(const Sym6& A) {
    Ldl6 F;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float d = A.m[sidx(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) d -= F.L[lidx(j, k)] * F.L[lidx(j, k)] * F.D[k];
        F.D[j] = d;
        float inv = 1.f / d;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            float s = A.m[sidx(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) s -= F.L[lidx(i, k)] * F.L[lidx(j, k)] * F.D[k];
            F.L[lidx(i, j)] = s * inv;
        }
    }
    return F;
}
MMS_HD S6 substitute6(const Ldl6& F, S6 bv) {
    float b[6] = {bv.a.x, bv.a.y, bv.a.z, bv.l.x, bv.l.y, bv.l.z};
    float y[6], x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= F.L[lidx(i, k)] * y[k];
        y[i] = s;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        float s = y[i] / F.D[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= F.L[lidx(k, i)] * x[k];
        x[i] = s;
    }
    return S6{V3{x[0], x[1], x[2]}, V3{x[3], x[4], x[5]}};
}
MMS_HD S6 solve6(const Sym6& A, S6 bv) { return substitute6(factor6(A), bv); }

// ---------------------------------------------------------------------------------------------
// contacts (oracle: contact_t / sphere_contacts / contact_fold / contact_force)
// ---------------------------------------------------------------------------------------------
struct Contact {
    float active;   // 0 / 1
    V3 xc, n, vrel;
    float kd, gn, ct;
};
MMS_HD Contact contact_none() { return Contact{0.f, V3{0, 0, 0}, V3{0, 0, 1}, V3{0, 0, 0}, 0.f, 0.f, 0.f}; }

// I^A += h P^T G P, p^A -= P^T (kd n - G vrel);  G = (gn - ct) n n^T + ct 1,  P = [-[xc]x | 1]:
// a rank-1 term along wn = (xc x n, n) plus the isotropic part h ct P^T P = h ct [[(x.x) 1 - x x^T, [x]x], [[x]x^T, 1]],
// whose 15 non-zero entries of the upper triangle are written out (three more rank-1 updates would cost 3 x 21 FMAs)
MMS_HD void contact_fold(const Contact& c, float h, Sym6& IA, S6& pA) {
    if (c.active == 0.f) return;
    S6 wn = S6{cross(c.xc, c.n), c.n};
    sym_rank1(IA, h * (c.gn - c.ct), wn);
    if (c.ct != 0.f) {
        const float t = h * c.ct, x = c.xc.x, y = c.xc.y, z = c.xc.z;
        IA.m[sidx(0, 0)] += t * (y * y + z * z); IA.m[sidx(0, 1)] += -t * x * y; IA.m[sidx(0, 2)] += -t * x * z;
        IA.m[sidx(1, 1)] += t * (x * x + z * z); IA.m[sidx(1, 2)] += -t * y * z; IA.m[sidx(2, 2)] += t * (x * x + y * y);
        IA.m[sidx(0, 4)] += -t * z; IA.m[sidx(0, 5)] += t * y;
        IA.m[sidx(1, 3)] += t * z;  IA.m[sidx(1, 5)] += -t * x;
        IA.m[sidx(2, 3)] += -t * y; IA.m[sidx(2, 4)] += t * x;
        IA.m[sidx(3, 3)] += t; IA.m[sidx(4, 4)] += t; IA.m[sidx(5, 5)] += t;
    }
    float vn = dot(c.n, c.vrel);
    V3 f0 = (c.kd - (c.gn - c.ct) * vn) * c.n - c.ct * c.vrel;
    pA.a = pA.a - cross(c.xc, f0);
    pA.l = pA.l - f0;
}
// f = kd n - G (vrel + h (a_lin + alpha x xc))
MMS_HD V3 contact_force(const Contact& c, float h, S6 acc) {
    if (c.active == 0.f) return V3{0, 0, 0};
    V3 u = c.vrel + h * (acc.l + cross(acc.a, c.xc));
    float un = dot(c.n, u);
    return (c.kd - (c.gn - c.ct) * un) * c.n - c.ct * u;
}

struct BoxPose { V3 pos; M3 R; V3 v, w, half; };

// sphere at xs (relative to the spatial origin at world position Ow) on a body with spatial velocity vb
MMS_HD Contact sphere_ground(float k, float cdamp, float mu, float slip_eps, float pen_ramp, float h, V3 Ow, V3 xs, float rad, S6 vb) {
    Contact c = contact_none();
    float d = rad - (Ow.z + xs.z);
    if (d > -kContactMargin) {
        V3 xc = V3{xs.x, xs.y, xs.z - rad};
        V3 vp = vb.l + cross(vb.a, xc);
        float w = ramp01(fmaxf(d, d - h * vp.z), pen_ramp);
        float gn = w * (h * k + cdamp);
        float fn = fmaxf(w * k * d - gn * vp.z, 0.f);
        if (w > 0.f) {
            c.active = 1.f;
            c.xc = xc;
            c.n = V3{0.f, 0.f, 1.f};
            c.kd = w * k * d;
            c.gn = gn;
            float vt = sqrtf(vp.x * vp.x + vp.y * vp.y);
            c.ct = mu * fn / fmaxf(vt, slip_eps);
            c.vrel = vp;
        }
    }
    return c;
}
MMS_HD Contact sphere_box(float k, float cdamp, float pen_ramp, float h, V3 Ow, V3 xs, float rad, S6 vb, const BoxPose& box,
                          float mu = 0.f, float slip_eps = 1.f) {
    Contact c = contact_none();
    V3 rel = Ow + xs - box.pos;
    V3 xb = mulT(box.R, rel);
    V3 q = V3{clampf(xb.x, -box.half.x, box.half.x), clampf(xb.y, -box.half.y, box.half.y), clampf(xb.z, -box.half.z, box.half.z)};
    bool inside = (q.x == xb.x) && (q.y == xb.y) && (q.z == xb.z);
    V3 nb = V3{0, 0, 0};
    float d;
    if (!inside) {
        V3 dl = xb - q;
        float dist = sqrtf(dot(dl, dl));
        d = rad - dist;
        if (d > -kContactMargin) nb = V3{dl.x / dist, dl.y / dist, dl.z / dist};
    } else {
        float mx = box.half.x - fabsf(xb.x), my = box.half.y - fabsf(xb.y), mz = box.half.z - fabsf(xb.z);
        int ax = 0;
        float best = mx;
        if (my < best) { best = my; ax = 1; }
        if (mz < best) { best = mz; ax = 2; }
        float sx = (xb.x >= 0.f) ? 1.f : -1.f, sy = (xb.y >= 0.f) ? 1.f : -1.f, sz = (xb.z >= 0.f) ? 1.f : -1.f;
        nb = V3{ax == 0 ? sx : 0.f, ax == 1 ? sy : 0.f, ax == 2 ? sz : 0.f};
        d = rad + best;
    }
    if (d > -kContactMargin) {
        V3 n = mul(box.R, nb);
        V3 xc = xs - rad * n;
        V3 vp = vb.l + cross(vb.a, xc);
        V3 rb = Ow + xc - box.pos;
        V3 vrel = vp - box.v - cross(box.w, rb);
        float w = ramp01(fmaxf(d, d - h * dot(n, vrel)), pen_ramp);
        float gn = w * (h * k + cdamp);
        if (w > 0.f) {
            c.active = 1.f;
            c.xc = xc;
            c.n = n;
            c.kd = w * k * d;
            c.gn = gn;
            c.ct = 0.f;
            if (mu > 0.f) {                                        // Coulomb friction, regularised like the ground's
                float vn = dot(n, vrel);
                float fn = fmaxf(w * k * d - gn * vn, 0.f);
                V3 vt = vrel - vn * n;
                c.ct = mu * fn / fmaxf(sqrtf(dot(vt, vt)), slip_eps);
            }
            c.vrel = vrel;
        }
    }
    return c;
}

// ---------------------------------------------------------------------------------------------
// per-leg constants (loaded once per lane from the device copy of mms_config)
// ---------------------------------------------------------------------------------------------
struct LegConst {
    V3 hip_pos, limb_dir, ankle_axis, limb_perp;   // limb_perp = ankle_axis x limb_dir
    float axis_dot_dir;                            // ankle_axis . limb_dir (0 for the ant)
    float lower[2], upper[2], init[2], gear[2];
};
MMS_HD LegConst load_leg_const(const mms_model* M, int l) {
    LegConst L;
    L.hip_pos = V3{M->hip_pos[l][0], M->hip_pos[l][1], M->hip_pos[l][2]};
    L.limb_dir = V3{M->limb_dir[l][0], M->limb_dir[l][1], M->limb_dir[l][2]};
    L.ankle_axis = V3{M->ankle_axis[l][0], M->ankle_axis[l][1], M->ankle_axis[l][2]};
    L.limb_perp = cross(L.ankle_axis, L.limb_dir);
    L.axis_dot_dir = dot(L.ankle_axis, L.limb_dir);
    for (int j = 0; j < 2; j++) {
        L.lower[j] = M->dof_lower[2 * l + j];
        L.upper[j] = M->dof_upper[2 * l + j];
        L.init[j] = M->dof_init[2 * l + j];
        L.gear[j] = M->gear[2 * l + j];
    }
    return L;
}

// replicated torso state + this lane's two joints
struct AntLane {
    V3 pos;
    float qx, qy, qz, qw;
    V3 vel, ang;
    float q[2], qd[2];
};

MMS_HD void quat_integrate(float& qx, float& qy, float& qz, float& qw, V3 w, float h) {
    float hx = 0.5f * h * w.x, hy = 0.5f * h * w.y, hz = 0.5f * h * w.z;
    float nx = qx + (hx * qw + hy * qz - hz * qy);
    float ny = qy + (hy * qw + hz * qx - hx * qz);
    float nz = qz + (hz * qw + hx * qy - hy * qx);
    float ns = qw - (hx * qx + hy * qy + hz * qz);
    float inv = 1.f / sqrtf(nx * nx + ny * ny + nz * nz + ns * ns);
    qx = nx * inv; qy = ny * inv; qz = nz * inv; qw = ns * inv;
}
MMS_HD void clamp_angvel(V3& w, float wmax) {
    float wn = sqrtf(dot(w, w));
    if (wn > wmax) { float s = wmax / wn; w = s * w; }
}

// sin / cos of a joint angle (|x| <= 2.5 rad: the joint ranges are +-0.7 and +-1.75 rad) by a half-angle
// polynomial: truncation error < 3e-9, i.e. fp32 rounding level; ~20 flops instead of a libm call each.
MMS_HD void sincos_joint(float x, float& s, float& c) {
    float y = 0.5f * clampf(x, -2.5f, 2.5f), y2 = y * y;
    float sy = y * (1.f + y2 * (-1.f / 6.f + y2 * (1.f / 120.f + y2 * (-1.f / 5040.f + y2 * (1.f / 362880.f + y2 * (-1.f / 39916800.f))))));
    float cy = 1.f + y2 * (-0.5f + y2 * (1.f / 24.f + y2 * (-1.f / 720.f + y2 * (1.f / 40320.f + y2 * (-1.f / 3628800.f + y2 * (1.f / 479001600.f))))));
    s = 2.f * sy * cy;
    c = 1.f - 2.f * sy * sy;
}
MMS_HD V3 rot_z(V3 v, float s, float c) { return V3{c * v.x - s * v.y, s * v.x + c * v.y, v.z}; }

// Broad phase: can any of this ant's spheres (all within `reach` of the torso centre) touch the box?
MMS_HD bool ant_near_box(V3 Ow, const BoxPose& box, float reach) {
    V3 xb = mulT(box.R, Ow - box.pos);
    float dx = fmaxf(fabsf(xb.x) - box.half.x, 0.f), dy = fmaxf(fabsf(xb.y) - box.half.y, 0.f), dz = fmaxf(fabsf(xb.z) - box.half.z, 0.f);
    return dx * dx + dy * dy + dz * dz < reach * reach;
}

// ground contact fold, n = (0,0,1), G = diag(ct, ct, gn): the sparse form of I^A += h P^T G P
MMS_HD void contact_fold_ground(const Contact& c, float h, Sym6& IA, S6& pA) {
    if (c.active == 0.f) return;
    float x = c.xc.x, y = c.xc.y, z = c.xc.z, ct = h * c.ct, gn = h * c.gn;
    IA.m[sidx(0, 0)] += ct * z * z + gn * y * y;
    IA.m[sidx(0, 1)] += -gn * x * y;
    IA.m[sidx(0, 2)] += -ct * x * z;
    IA.m[sidx(1, 1)] += ct * z * z + gn * x * x;
    IA.m[sidx(1, 2)] += -ct * y * z;
    IA.m[sidx(2, 2)] += ct * (x * x + y * y);
    IA.m[sidx(1, 3)] += ct * z;  IA.m[sidx(2, 3)] += -ct * y;      // column of e_x: ct (0, z, -y)
    IA.m[sidx(0, 4)] += -ct * z; IA.m[sidx(2, 4)] += ct * x;       // column of e_y: ct (-z, 0, x)
    IA.m[sidx(0, 5)] += gn * y;  IA.m[sidx(1, 5)] += -gn * x;      // column of e_z: gn (y, -x, 0)
    IA.m[sidx(3, 3)] += ct; IA.m[sidx(4, 4)] += ct; IA.m[sidx(5, 5)] += gn;
    V3 f0 = V3{-c.ct * c.vrel.x, -c.ct * c.vrel.y, c.kd - c.gn * c.vrel.z};
    pA.a = pA.a - cross(c.xc, f0);
    pA.l = pA.l - f0;
}
// box contact fold.  Frictionless (rank 1): I^A += h gn wn wn^T, p^A -= (kd - gn vn) wn,  wn = (xc x n, n); with friction
// (c.ct != 0, model.antbox_mu > 0) the general form.
MMS_HD void contact_fold_box(const Contact& c, float h, Sym6& IA, S6& pA) {
    if (c.active == 0.f) return;
    if (c.ct != 0.f) { contact_fold(c, h, IA, pA); return; }
    S6 wn = S6{cross(c.xc, c.n), c.n};
    sym_rank1(IA, h * c.gn, wn);
    pA = pA + (-(c.kd - c.gn * dot(c.n, c.vrel))) * wn;
}
// reaction of one (re-evaluated) box contact on the box, given the body's acceleration:
// f = (kd - gn vn) - h gn (wn . a);  wrench -= f ((O + xc - box) x n, n)
MMS_HD void box_reaction(const Contact& c, float h, V3 Ow, const BoxPose& box, S6 acc, S6& w) {
    if (c.active == 0.f) return;
    if (c.ct != 0.f) {                                             // with friction: the full contact force, tangential part included
        V3 f = contact_force(c, h, acc);
        w.a = w.a - cross(Ow + c.xc - box.pos, f);
        w.l = w.l - f;
        return;
    }
    S6 wn = S6{cross(c.xc, c.n), c.n};
    float f = (c.kd - c.gn * dot(c.n, c.vrel)) - (h * c.gn) * dot(wn, acc);
    w.a = w.a - f * cross(Ow + c.xc - box.pos, c.n);
    w.l = w.l - f * c.n;
}

// Compiler fences used by the step kernel to stop it from keeping recomputable values alive across the
// quad reduction and the root solve (register pressure decides how many waves are resident at once).
#if defined(__HIP_DEVICE_COMPILE__)
#define MMS_REG_FENCE(x) asm volatile("" : "+v"(x))
#define MMS_MEM_FENCE() asm volatile("" ::: "memory")
#else
#define MMS_REG_FENCE(x) ((void)0)
#define MMS_MEM_FENCE() ((void)0)
#endif

// kinematics of one leg chain; a pure function of (S, L), evaluated in BOTH passes instead of being kept alive
struct LegKin {
    V3 J1, J2, tip, ul, uf;
    S6 s1, s2, vl, vf, c1, c2;
};
// sc = (sin q1, cos q1, sin q2, cos q2)
MMS_HD LegKin leg_kinematics(const mms_model* M, const LegConst& L, const AntLane& S, const M3& Rt, const float sc[4]) {
    LegKin K;
    S6 v0 = S6{S.ang, S.vel};
    const float s1q = sc[0], c1q = sc[1], s2q = sc[2], c2q = sc[3];
    // R_leg = R_t Rz(q1), R_foot = R_leg Rot(ankle_axis, q2); only the vectors that are needed
    V3 a1 = Rt.c2;
    K.J1 = mul(Rt, L.hip_pos);
    K.ul = mul(Rt, rot_z(L.limb_dir, s1q, c1q));
    V3 a2 = mul(Rt, rot_z(L.ankle_axis, s1q, c1q));
    V3 pw = mul(Rt, rot_z(L.limb_perp, s1q, c1q));
    K.J2 = K.J1 + M->leg_len * K.ul;
    K.uf = c2q * K.ul + s2q * pw + ((1.f - c2q) * L.axis_dot_dir) * a2;    // Rodrigues applied to limb_dir
    K.tip = K.J2 + M->foot_len * K.uf;
    K.s1 = S6{a1, cross(K.J1, a1)};
    K.s2 = S6{a2, cross(K.J2, a2)};
    S6 sq1 = S.qd[0] * K.s1;
    K.vl = v0 + sq1;
    K.c1 = cross_motion(v0, sq1);
    S6 sq2 = S.qd[1] * K.s2;
    K.vf = K.vl + sq2;
    K.c2 = cross_motion(K.vl, sq2);
    return K;
}

// What a leg lane keeps between the inward and the outward pass: 20 values + the broad-phase flag
struct LegPass {
    S6 U1, U2;
    float D1, D2, u1, u2;
    float sc[4];          // sin / cos of the two joint angles (the rest of the kinematics is evaluated again)
    bool near_box;        // broad phase: some sphere of this ant may reach the box
    bool touch_box;       // ... and one of THIS lane's spheres does: only then the outward pass evaluates the reactions on the box
};
// extra state for the foot force sensors (OneAnt only)
struct SensorPass { Contact tip_g, tip_b; M3 Rf; V3 J2; };
// Where a lane parks the joint axes and velocity-product terms (s1, s2, c1, c2: 24 floats) between its two passes instead
// of evaluating the kinematics again: six 16-B words at `base[k * stride]` (LDS in the step kernel, lane-interleaved so that
// the accesses are conflict free).  base == nullptr: evaluate again (the host emulation, which has no such scratch).
struct KinPark { float* base; int stride; };
// Per-lane slice of an ant's physical domain-randomisation block (mms.h: MMS_DR_FLOATS; oracle ant_substep): mass scales of
// the torso and of this lane's leg / foot (inertia scales with the mass), damping scale and limit offsets of its two joints.
// Lives in LDS in the step kernel and is read at the point of use.
struct LegDR { float m_torso, m_leg, m_foot, damp[2], lo[2], hi[2]; };
MMS_HD LegDR load_leg_dr(const float* ant_block, int leg) {
    LegDR d;
    d.m_torso = ant_block[0]; d.m_leg = ant_block[1 + leg]; d.m_foot = ant_block[5 + leg];
    for (int j = 0; j < 2; j++) {
        d.damp[j] = ant_block[9 + 2 * leg + j]; d.lo[j] = ant_block[17 + 2 * leg + j]; d.hi[j] = ant_block[25 + 2 * leg + j];
    }
    return d;
}
MMS_HD void park_store(const KinPark& pk, int k, float a, float b, float c, float d) {
    float* p = pk.base + (size_t)k * pk.stride;
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
}
MMS_HD void park_load(const KinPark& pk, int k, float& a, float& b, float& c, float& d) {
    const float* p = pk.base + (size_t)k * pk.stride;
    a = p[0]; b = p[1]; c = p[2]; d = p[3];
}

// joint torque with linearly-implicit damping and limits: returns tau, adds to De
MMS_HD float joint_tau(const mms_model* M, float h, float q, float qd, float lo, float hi, float motor, float& De, float damping) {
    float t = motor - damping * qd;
    De = M->armature + h * damping;
    float ehi = q - hi, elo = lo - q;
    float whi = ramp01(fmaxf(ehi, ehi + h * qd), M->limit_ramp), wlo = ramp01(fmaxf(elo, elo - h * qd), M->limit_ramp);
    if (whi > 0.f) {
        float gl = whi * (h * M->limit_k + M->limit_c);
        t += -whi * M->limit_k * ehi - gl * qd; De += h * gl;
    } else if (wlo > 0.f) {
        float gl = wlo * (h * M->limit_k + M->limit_c);
        t += wlo * M->limit_k * elo - gl * qd; De += h * gl;
    }
    return t;
}

MMS_HD float ant_reach(const mms_model* M, const LegConst& L) {
    // hip offset + leg + foot + sphere radius + activation margin
    float r = sqrtf(dot(L.hip_pos, L.hip_pos)) + M->leg_len + M->foot_len + M->limb_radius + kContactMargin;
    return fmaxf(r, M->torso_radius + kContactMargin);
}

// Phase A (inward pass of one leg chain).  Returns this lane's contribution (Ia, pa) to the torso's
// articulated inertia; lane l == 0 also adds the torso body itself and the torso sphere contacts.
// In two parts: `leg_inward_open` is everything that does not depend on the box (kinematics, the foot body's inertia and bias
// force, the tip's ground contact), `leg_inward_close` the rest.  The step kernel puts the barrier behind which this substep's box
// pose is valid BETWEEN them, so that the ant waves work through the opening while the box lanes finish the previous substep's
// serial tail; `leg_inward` is the two back to back (host builds, one-wave layouts).
struct LegInward { M3 Rt; LegKin K; Sym6 IAf; S6 pAf; };

template <bool SENSORS, bool DR = false>
MMS_HD void leg_inward_open(const mms_model* M, const LegConst& L, float h, const AntLane& S, LegPass& P, SensorPass* SP, LegInward& W,
                            const KinPark& park = KinPark{nullptr, 0}, const LegDR* dr = nullptr) {
    W.Rt = quat_to_mat(S.qx, S.qy, S.qz, S.qw);
    V3 Ow = S.pos;
    sincos_joint(S.q[0], P.sc[0], P.sc[1]);
    sincos_joint(S.q[1], P.sc[2], P.sc[3]);
    W.K = leg_kinematics(M, L, S, W.Rt, P.sc);
    const LegKin& K = W.K;
    if (park.base) {
        park_store(park, 0, K.s1.a.x, K.s1.a.y, K.s1.a.z, K.s1.l.x);
        park_store(park, 1, K.s1.l.y, K.s1.l.z, K.s2.a.x, K.s2.a.y);
        park_store(park, 2, K.s2.a.z, K.s2.l.x, K.s2.l.y, K.s2.l.z);
        park_store(park, 3, K.c1.a.x, K.c1.a.y, K.c1.a.z, K.c1.l.x);
        park_store(park, 4, K.c1.l.y, K.c1.l.z, K.c2.a.x, K.c2.a.y);
        park_store(park, 5, K.c2.a.z, K.c2.l.x, K.c2.l.y, K.c2.l.z);
    }
    // ---- foot body: inertia, bias force, the tip's ground contact ------------------------------------------------
    V3 cf = K.J2 + (0.5f * M->foot_len) * K.uf;
    const float mf = DR ? dr->m_foot : 1.f;
    spatial_inertia_axisym(M->foot_mass * mf, cf, K.uf, M->foot_ia * mf, M->foot_it * mf, W.IAf);
    W.pAf = bias_force_axisym(K.vf, M->foot_mass * mf, cf, K.uf, M->foot_ia * mf, M->foot_it * mf, M->gravity);
    Contact g = sphere_ground(M->gnd_k, M->gnd_c, M->gnd_mu, M->slip_eps, M->pen_ramp, h, Ow, K.tip, M->limb_radius, K.vf);
    contact_fold_ground(g, h, W.IAf, W.pAf);
    if (SENSORS) { SP->tip_g = g; SP->tip_b = contact_none(); }
}

template <bool SENSORS, bool DR = false>
MMS_HD void leg_inward_close(const mms_model* M, const LegConst& L, float h, const AntLane& S, int leg, float tau1, float tau2,
                             bool has_box, const BoxPose& box, LegPass& P, SensorPass* SP, Sym6& IA0, S6& pA0, LegInward& W,
                             const LegDR* dr = nullptr) {
    const M3& Rt = W.Rt;
    const LegKin& K = W.K;
    Sym6& IAf = W.IAf;
    S6& pAf = W.pAf;
    V3 Ow = S.pos;
    S6 v0 = S6{S.ang, S.vel};
    P.near_box = has_box && ant_near_box(Ow, box, ant_reach(M, L));
    P.touch_box = false;
    // ---- foot body: the tip's box contact, joint 2 ---------------------------------------------------------------
    if (MMS_UNLIKELY(P.near_box)) {
        Contact b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, K.tip, M->limb_radius, K.vf, box, M->antbox_mu, M->slip_eps);
        contact_fold_box(b, h, IAf, pAf);
        P.touch_box = P.touch_box || b.active != 0.f;
        if (SENSORS) SP->tip_b = b;
    }
    float De1, De2;
    float t2 = DR ? joint_tau(M, h, S.q[1], S.qd[1], L.lower[1] + dr->lo[1], L.upper[1] + dr->hi[1], tau2, De2, M->joint_damping * dr->damp[1])
                  : joint_tau(M, h, S.q[1], S.qd[1], L.lower[1], L.upper[1], tau2, De2, M->joint_damping);
    P.U2 = sym_mul(IAf, K.s2);
    P.D2 = De2 + dot(K.s2, P.U2);
    P.u2 = t2 - dot(K.s2, pAf);
    S6 pa_f;
    {
        float invD = 1.f / P.D2;
        sym_rank1(IAf, -invD, P.U2);                       // Ia = IA - U U^T / D
        S6 Iac = sym_mul(IAf, K.c2);
        pa_f = pAf + Iac + (P.u2 * invD) * P.U2;
    }
    // ---- leg body: inertia, bias force, hip / knee contacts, joint 1 -------------------------------------------
    Sym6 IAl;
    V3 cl = K.J1 + (0.5f * M->leg_len) * K.ul;
    const float ml = DR ? dr->m_leg : 1.f;
    spatial_inertia_axisym(M->leg_mass * ml, cl, K.ul, M->leg_ia * ml, M->leg_it * ml, IAl);
    S6 pAl = bias_force_axisym(K.vl, M->leg_mass * ml, cl, K.ul, M->leg_ia * ml, M->leg_it * ml, M->gravity);
    sym_add(IAl, IAf);                                     // the foot's articulated inertia joins before the leg's contacts are
    pAl = pAl + pa_f;                                      // folded in: 27 fewer live values at the register-pressure peak
    {
        Contact g = sphere_ground(M->gnd_k, M->gnd_c, M->gnd_mu, M->slip_eps, M->pen_ramp, h, Ow, K.J1, M->limb_radius, K.vl);
        contact_fold_ground(g, h, IAl, pAl);
        g = sphere_ground(M->gnd_k, M->gnd_c, M->gnd_mu, M->slip_eps, M->pen_ramp, h, Ow, K.J2, M->limb_radius, K.vl);
        contact_fold_ground(g, h, IAl, pAl);
        if (MMS_UNLIKELY(P.near_box)) {
            Contact b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, K.J1, M->limb_radius, K.vl, box, M->antbox_mu, M->slip_eps);
            contact_fold_box(b, h, IAl, pAl);
            P.touch_box = P.touch_box || b.active != 0.f;
            b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, K.J2, M->limb_radius, K.vl, box, M->antbox_mu, M->slip_eps);
            contact_fold_box(b, h, IAl, pAl);
            P.touch_box = P.touch_box || b.active != 0.f;
        }
    }
    float t1 = DR ? joint_tau(M, h, S.q[0], S.qd[0], L.lower[0] + dr->lo[0], L.upper[0] + dr->hi[0], tau1, De1, M->joint_damping * dr->damp[0])
                  : joint_tau(M, h, S.q[0], S.qd[0], L.lower[0], L.upper[0], tau1, De1, M->joint_damping);
    P.U1 = sym_mul(IAl, K.s1);
    P.D1 = De1 + dot(K.s1, P.U1);
    P.u1 = t1 - dot(K.s1, pAl);
    {
        float invD = 1.f / P.D1;
        sym_rank1(IAl, -invD, P.U1);
        S6 Iac = sym_mul(IAl, K.c1);
        IA0 = IAl;
        pA0 = pAl + Iac + (P.u1 * invD) * P.U1;
    }
    if (SENSORS) {
        M3 Rl = mul(Rt, axis_angle_to_mat(V3{0.f, 0.f, 1.f}, S.q[0]));
        SP->Rf = mul(Rl, axis_angle_to_mat(L.ankle_axis, S.q[1]));
        SP->J2 = K.J2;
    }
    // ---- the torso body itself (lane 0 of the quad adds it once) ------------------------------------------------
    // Its COM is the frame origin: the spatial inertia is block diagonal, (it 1 + (ia - it) u u^T, m 1), and the bias force is
    // (w x Ic w, m w x v + m g z) -- the general routines with c = 0, minus the terms that are zero by construction.
    if (leg == 0) {
        const V3 u = Rt.c2;
        const float mt = DR ? dr->m_torso : 1.f;
        const float m = M->torso_mass * mt, it = M->torso_ixx * mt, d = (M->torso_izz - M->torso_ixx) * mt;
        IA0.m[sidx(0, 0)] += it + d * u.x * u.x; IA0.m[sidx(0, 1)] += d * u.x * u.y; IA0.m[sidx(0, 2)] += d * u.x * u.z;
        IA0.m[sidx(1, 1)] += it + d * u.y * u.y; IA0.m[sidx(1, 2)] += d * u.y * u.z; IA0.m[sidx(2, 2)] += it + d * u.z * u.z;
        IA0.m[sidx(3, 3)] += m; IA0.m[sidx(4, 4)] += m; IA0.m[sidx(5, 5)] += m;
        const V3 n = it * v0.a + (d * dot(u, v0.a)) * u;          // Ic w
        const V3 f = m * v0.l;
        pA0.a = pA0.a + cross(v0.a, n);
        pA0.l = pA0.l + (cross(v0.a, f) - V3{0.f, 0.f, -m * M->gravity});
        V3 zero = V3{0, 0, 0};
        Contact g = sphere_ground(M->gnd_k, M->gnd_c, M->gnd_mu, M->slip_eps, M->pen_ramp, h, Ow, zero, M->torso_radius, v0);
        contact_fold_ground(g, h, IA0, pA0);
        if (MMS_UNLIKELY(P.near_box)) {
            Contact b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, zero, M->torso_radius, v0, box, M->antbox_mu, M->slip_eps);
            contact_fold_box(b, h, IA0, pA0);
            P.touch_box = P.touch_box || b.active != 0.f;
        }
    }
}

template <bool SENSORS, bool DR = false>
MMS_HD void leg_inward(const mms_model* M, const LegConst& L, float h, const AntLane& S, int leg, float tau1, float tau2,
                       bool has_box, const BoxPose& box, LegPass& P, SensorPass* SP, Sym6& IA0, S6& pA0, const KinPark& park = KinPark{nullptr, 0},
                       const LegDR* dr = nullptr) {
    LegInward W;
    leg_inward_open<SENSORS, DR>(M, L, h, S, P, SP, W, park, dr);
    leg_inward_close<SENSORS, DR>(M, L, h, S, leg, tau1, tau2, has_box, box, P, SP, IA0, pA0, W, dr);
}

// Phase B (after the quad reduction): root solve, outward pass, contact forces, integration.
// `wrench` returns this lane's reaction on the box; `sens` the foot sensor (force, torque) in the foot frame.
// The joint axes and velocity products come back from the lane's parking space (or are evaluated again from the unchanged
// state: bit-identical), the (rare) box contacts are evaluated again: only 20 values per lane stay in registers across the
// reduction and the solve (the register count decides whether the whole 4096-env grid is resident at once).
template <bool SENSORS>
MMS_HD void leg_outward(const mms_model* M, const LegConst& L, float h, AntLane& S, int leg, const BoxPose& box, const LegPass& P,
                        const SensorPass* SP, const Sym6& IA0, S6 pA0, S6& wrench, float* sens, const KinPark& park = KinPark{nullptr, 0}) {
    S6 rhs = S6{V3{-pA0.a.x, -pA0.a.y, -pA0.a.z}, V3{-pA0.l.x, -pA0.l.y, -pA0.l.z}};
    S6 a0 = solve6(IA0, rhs);
    MMS_REG_FENCE(S.pos.x); MMS_REG_FENCE(S.pos.y); MMS_REG_FENCE(S.pos.z);
    MMS_REG_FENCE(S.qx); MMS_REG_FENCE(S.qy); MMS_REG_FENCE(S.qz); MMS_REG_FENCE(S.qw);
    MMS_REG_FENCE(S.vel.x); MMS_REG_FENCE(S.vel.y); MMS_REG_FENCE(S.vel.z);
    MMS_REG_FENCE(S.ang.x); MMS_REG_FENCE(S.ang.y); MMS_REG_FENCE(S.ang.z);
    MMS_REG_FENCE(S.q[0]); MMS_REG_FENCE(S.q[1]); MMS_REG_FENCE(S.qd[0]); MMS_REG_FENCE(S.qd[1]);
    MMS_MEM_FENCE();
    S6 s1, s2, c1, c2;
    if (park.base) {
        park_load(park, 0, s1.a.x, s1.a.y, s1.a.z, s1.l.x);
        park_load(park, 1, s1.l.y, s1.l.z, s2.a.x, s2.a.y);
        park_load(park, 2, s2.a.z, s2.l.x, s2.l.y, s2.l.z);
        park_load(park, 3, c1.a.x, c1.a.y, c1.a.z, c1.l.x);
        park_load(park, 4, c1.l.y, c1.l.z, c2.a.x, c2.a.y);
        park_load(park, 5, c2.a.z, c2.l.x, c2.l.y, c2.l.z);
    } else {
        M3 Rt = quat_to_mat(S.qx, S.qy, S.qz, S.qw);
        LegKin K = leg_kinematics(M, L, S, Rt, P.sc);
        s1 = K.s1; s2 = K.s2; c1 = K.c1; c2 = K.c2;
    }
    V3 Ow = S.pos;
    S6 al = a0 + c1;
    float qdd1 = (P.u1 - dot(P.U1, al)) / P.D1;
    al = al + qdd1 * s1;
    S6 af = al + c2;
    float qdd2 = (P.u2 - dot(P.U2, af)) / P.D2;
    af = af + qdd2 * s2;
    wrench = S6{V3{0, 0, 0}, V3{0, 0, 0}};
    // Only a lane that HAD an active box contact in the inward pass (same state, same box pose: the same contacts are active here)
    // evaluates them again; the broad phase alone is true for every ant standing around the box, i.e. for every wave of TenAnt.
    if (MMS_UNLIKELY(P.touch_box)) {
        M3 Rt = quat_to_mat(S.qx, S.qy, S.qz, S.qw);          // the full kinematics again for the contact points
        LegKin K = leg_kinematics(M, L, S, Rt, P.sc);
        S6 v0 = S6{S.ang, S.vel};
        Contact b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, K.J1, M->limb_radius, K.vl, box, M->antbox_mu, M->slip_eps);
        box_reaction(b, h, Ow, box, al, wrench);
        b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, K.J2, M->limb_radius, K.vl, box, M->antbox_mu, M->slip_eps);
        box_reaction(b, h, Ow, box, al, wrench);
        b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, K.tip, M->limb_radius, K.vf, box, M->antbox_mu, M->slip_eps);
        box_reaction(b, h, Ow, box, af, wrench);
        if (leg == 0) {
            b = sphere_box(M->antbox_k, M->antbox_c, M->pen_ramp, h, Ow, V3{0, 0, 0}, M->torso_radius, v0, box, M->antbox_mu, M->slip_eps);
            box_reaction(b, h, Ow, box, a0, wrench);
        }
    }
    if (SENSORS) {
        V3 f_tip_g = contact_force(SP->tip_g, h, af), f_tip_b = contact_force(SP->tip_b, h, af);
        V3 F = V3{0, 0, 0}, T = V3{0, 0, 0};
        if (SP->tip_g.active != 0.f) { F = F + f_tip_g; T = T + cross(SP->tip_g.xc - SP->J2, f_tip_g); }
        if (SP->tip_b.active != 0.f) { F = F + f_tip_b; T = T + cross(SP->tip_b.xc - SP->J2, f_tip_b); }
        V3 Fl = mulT(SP->Rf, F), Tl = mulT(SP->Rf, T);
        sens[0] = Fl.x; sens[1] = Fl.y; sens[2] = Fl.z; sens[3] = Tl.x; sens[4] = Tl.y; sens[5] = Tl.z;
    }
    // integrate (semi-implicit Euler); S still holds the pre-step velocities here
    V3 wxv = cross(S.ang, S.vel);
    S.qd[0] += h * qdd1; S.q[0] += h * S.qd[0];
    S.qd[1] += h * qdd2; S.q[1] += h * S.qd[1];
    S.vel = S.vel + h * (a0.l + wxv);
    S.ang = S.ang + h * a0.a;
    clamp_angvel(S.ang, kMaxAngVel);
    S.pos = S.pos + h * S.vel;
    quat_integrate(S.qx, S.qy, S.qz, S.qw, S.ang, h);
}

// ---------------------------------------------------------------------------------------------
// box: one corner lane's contribution, then the 6x6 solve (after the 8-lane reduction)
// ---------------------------------------------------------------------------------------------
struct RigidState { V3 pos; float qx, qy, qz, qw; V3 vel, ang; };

// One corner's ground contact.  The ground normal is z and the box is frictionless, so the implicit term
// h gn w w^T, w = (xc x z, z) = (y, -x, 0, 0, 0, 1), has six distinct non-zero entries and the right-hand side three:
//   t[0] = g y^2   t[1] = -g x y   t[2] = g x^2   t[3] = g y   t[4] = -g x   t[5] = g        (g = h * gn)
//   t[6] = fz y    t[7] = -fz x    t[8] = fz                                                 (fz = kd - gn vz)
// Only these nine values are reduced over the eight corner lanes.
struct BoxCorner { float t[9]; };
MMS_HD BoxCorner box_corner(const mms_model* M, float h, const RigidState& B, const M3& R, int corner) {
    BoxCorner o;
#pragma unroll
    for (int k = 0; k < 9; k++) o.t[k] = 0.f;
    V3 loc = V3{(corner & 1 ? 1.f : -1.f) * M->box_half[0], (corner & 2 ? 1.f : -1.f) * M->box_half[1],
                (corner & 4 ? 1.f : -1.f) * M->box_half[2]};
    V3 xc = mul(R, loc);
    float d = -(B.pos.z + xc.z);
    if (d <= -kContactMargin) return o;
    V3 vp = B.vel + cross(B.ang, xc);
    float w = ramp01(fmaxf(d, d - h * vp.z), M->pen_ramp);
    if (!(w > 0.f)) return o;
    float gn = w * (h * M->boxgnd_k + M->boxgnd_c);
    float g = h * gn, fz = w * M->boxgnd_k * d - gn * vp.z;
    o.t[0] = g * xc.y * xc.y; o.t[1] = -g * xc.x * xc.y; o.t[2] = g * xc.x * xc.x;
    o.t[3] = g * xc.y; o.t[4] = -g * xc.x; o.t[5] = g;
    o.t[6] = fz * xc.y; o.t[7] = -fz * xc.x; o.t[8] = fz;
    return o;
}
// Box update from the reduced corner terms and the ants' reaction wrench (torque about the COM, force).
// (I_w + C) alpha + g a_z = tau,  g^T alpha + (m + c55) a_z = f_z,  m a_xy = f_xy: the 6x6 system reduces to a
// symmetric 3x3 one for alpha by eliminating a_z.
MMS_HD void box_finish(const mms_model* M, float h, RigidState& B, const M3& R, const BoxCorner& c, S6 wrench) {
    // I_w = R diag(I) R^T
    V3 d = V3{M->box_inertia[0], M->box_inertia[1], M->box_inertia[2]};
    float ixx = d.x * R.c0.x * R.c0.x + d.y * R.c1.x * R.c1.x + d.z * R.c2.x * R.c2.x;
    float ixy = d.x * R.c0.x * R.c0.y + d.y * R.c1.x * R.c1.y + d.z * R.c2.x * R.c2.y;
    float ixz = d.x * R.c0.x * R.c0.z + d.y * R.c1.x * R.c1.z + d.z * R.c2.x * R.c2.z;
    float iyy = d.x * R.c0.y * R.c0.y + d.y * R.c1.y * R.c1.y + d.z * R.c2.y * R.c2.y;
    float iyz = d.x * R.c0.y * R.c0.z + d.y * R.c1.y * R.c1.z + d.z * R.c2.y * R.c2.z;
    float izz = d.x * R.c0.z * R.c0.z + d.y * R.c1.z * R.c1.z + d.z * R.c2.z * R.c2.z;
    V3 Iw_w = V3{ixx * B.ang.x + ixy * B.ang.y + ixz * B.ang.z, ixy * B.ang.x + iyy * B.ang.y + iyz * B.ang.z,
                 ixz * B.ang.x + iyz * B.ang.y + izz * B.ang.z};
    V3 gyro = cross(B.ang, Iw_w);
    V3 tau = wrench.a - gyro + V3{c.t[6], c.t[7], 0.f};
    V3 f = wrench.l + V3{0.f, 0.f, c.t[8] - M->box_mass * M->gravity};
    float m55 = M->box_mass + c.t[5];
    float inv55 = 1.f / m55;
    float g0 = c.t[3], g1 = c.t[4];
    // Schur complement on alpha
    float a00 = ixx + c.t[0] - g0 * g0 * inv55, a01 = ixy + c.t[1] - g0 * g1 * inv55, a02 = ixz;
    float a11 = iyy + c.t[2] - g1 * g1 * inv55, a12 = iyz, a22 = izz;
    float r0 = tau.x - g0 * f.z * inv55, r1 = tau.y - g1 * f.z * inv55, r2 = tau.z;
    // 3x3 LDL^T
    float d0 = a00, l10 = a01 / d0, l20 = a02 / d0;
    float d1 = a11 - l10 * l10 * d0, l21 = (a12 - l20 * l10 * d0) / d1;
    float d2 = a22 - l20 * l20 * d0 - l21 * l21 * d1;
    float y0 = r0, y1 = r1 - l10 * y0, y2 = r2 - l20 * y0 - l21 * y1;
    float az_ = y2 / d2;
    float ay_ = y1 / d1 - l21 * az_;
    float ax_ = y0 / d0 - l10 * ay_ - l20 * az_;
    V3 alpha = V3{ax_, ay_, az_};
    float inv_m = 1.f / M->box_mass;
    V3 acc = V3{f.x * inv_m, f.y * inv_m, (f.z - g0 * alpha.x - g1 * alpha.y) * inv55};
    B.ang = B.ang + h * alpha;
    B.vel = B.vel + h * acc;
    clamp_angvel(B.ang, kMaxAngVel);
    B.pos = B.pos + h * B.vel;
    quat_integrate(B.qx, B.qy, B.qz, B.qw, B.ang, h);
}

// The same with Coulomb friction between the box and the ground (model.boxgnd_mu > 0): the tangential terms couple all six
// accelerations, so a corner contributes a full (sparse-filled) 6x6 + 6-vector and the box is solved like an ant's torso.
struct BoxCornerF { Sym6 IA; S6 pA; };
MMS_HD BoxCornerF box_corner_friction(const mms_model* M, float h, const RigidState& B, const M3& R, int corner) {
    BoxCornerF o;
    sym_zero(o.IA);
    o.pA = S6{V3{0, 0, 0}, V3{0, 0, 0}};
    V3 loc = V3{(corner & 1 ? 1.f : -1.f) * M->box_half[0], (corner & 2 ? 1.f : -1.f) * M->box_half[1],
                (corner & 4 ? 1.f : -1.f) * M->box_half[2]};
    V3 xc = mul(R, loc);
    float d = -(B.pos.z + xc.z);
    if (d <= -kContactMargin) return o;
    V3 vp = B.vel + cross(B.ang, xc);
    float w = ramp01(fmaxf(d, d - h * vp.z), M->pen_ramp);
    if (!(w > 0.f)) return o;
    Contact c = contact_none();
    c.active = 1.f;
    c.xc = xc;
    c.n = V3{0.f, 0.f, 1.f};
    c.gn = w * (h * M->boxgnd_k + M->boxgnd_c);
    c.kd = w * M->boxgnd_k * d;
    float fn = fmaxf(c.kd - c.gn * vp.z, 0.f);
    float vt = sqrtf(vp.x * vp.x + vp.y * vp.y);
    c.ct = M->boxgnd_mu * fn / fmaxf(vt, M->slip_eps);
    c.vrel = vp;
    contact_fold_ground(c, h, o.IA, o.pA);
    return o;
}
// ... in two parts: everything that does not need the ants' reaction (matrix, its factorisation, the part of the right-hand side
// made of gravity, the gyroscopic term and the contacts) and the rest (add the reaction, substitute, integrate)
struct BoxFactorF { Ldl6 F; S6 rhs0; };
MMS_HD BoxFactorF box_factor_friction(const mms_model* M, const RigidState& B, const M3& R, const BoxCornerF& c) {
    Sym6 A = c.IA;
    V3 d = V3{M->box_inertia[0], M->box_inertia[1], M->box_inertia[2]};
    float ixx = d.x * R.c0.x * R.c0.x + d.y * R.c1.x * R.c1.x + d.z * R.c2.x * R.c2.x;
    float ixy = d.x * R.c0.x * R.c0.y + d.y * R.c1.x * R.c1.y + d.z * R.c2.x * R.c2.y;
    float ixz = d.x * R.c0.x * R.c0.z + d.y * R.c1.x * R.c1.z + d.z * R.c2.x * R.c2.z;
    float iyy = d.x * R.c0.y * R.c0.y + d.y * R.c1.y * R.c1.y + d.z * R.c2.y * R.c2.y;
    float iyz = d.x * R.c0.y * R.c0.z + d.y * R.c1.y * R.c1.z + d.z * R.c2.y * R.c2.z;
    float izz = d.x * R.c0.z * R.c0.z + d.y * R.c1.z * R.c1.z + d.z * R.c2.z * R.c2.z;
    A.m[sidx(0, 0)] += ixx; A.m[sidx(0, 1)] += ixy; A.m[sidx(0, 2)] += ixz;
    A.m[sidx(1, 1)] += iyy; A.m[sidx(1, 2)] += iyz; A.m[sidx(2, 2)] += izz;
    A.m[sidx(3, 3)] += M->box_mass; A.m[sidx(4, 4)] += M->box_mass; A.m[sidx(5, 5)] += M->box_mass;
    V3 Iw_w = V3{ixx * B.ang.x + ixy * B.ang.y + ixz * B.ang.z, ixy * B.ang.x + iyy * B.ang.y + iyz * B.ang.z,
                 ixz * B.ang.x + iyz * B.ang.y + izz * B.ang.z};
    V3 gyro = cross(B.ang, Iw_w);
    BoxFactorF o;
    o.F = factor6(A);
    o.rhs0 = S6{V3{0.f, 0.f, 0.f} - gyro - c.pA.a, V3{0.f, 0.f, -M->box_mass * M->gravity} - c.pA.l};
    return o;
}
MMS_HD void box_solve_friction(float h, RigidState& B, const BoxFactorF& f, S6 wrench) {
    S6 rhs = S6{wrench.a + f.rhs0.a, wrench.l + f.rhs0.l};
    S6 acc = substitute6(f.F, rhs);
    B.ang = B.ang + h * acc.a;
    B.vel = B.vel + h * acc.l;
    clamp_angvel(B.ang, kMaxAngVel);
    B.pos = B.pos + h * B.vel;
    quat_integrate(B.qx, B.qy, B.qz, B.qw, B.ang, h);
}
MMS_HD void box_finish_friction(const mms_model* M, float h, RigidState& B, const M3& R, const BoxCornerF& c, S6 wrench) {
    box_solve_friction(h, B, box_factor_friction(M, B, R, c), wrench);
}

// ---------------------------------------------------------------------------------------------
// helicopter (one lane per helicopter; oracle heli_substep)
// ---------------------------------------------------------------------------------------------
MMS_HD void heli_substep(const mms_model* M, float h, RigidState& B, V3 thr0, V3 thr1) {
    M3 R = quat_to_mat(B.qx, B.qy, B.qz, B.qw);
    V3 c = mul(R, V3{0.f, 0.f, M->heli_com_z});
    Sym6 A;
    spatial_inertia_diag(M->heli_mass, c, R, V3{M->heli_inertia[0], M->heli_inertia[1], M->heli_inertia[2]}, A);
    S6 v0 = S6{B.ang, B.vel};
    S6 p = bias_force(A, v0, M->heli_mass, c, M->gravity);
    V3 thr[2] = {thr0, thr1};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        V3 x = mul(R, V3{0.f, 0.f, M->heli_rotor_z[r]});
        V3 f = mul(R, thr[r]);
        p.a = p.a - cross(x, f);
        p.l = p.l - f;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        V3 loc = V3{(k & 1 ? 1.f : -1.f) * M->heli_half, (k & 2 ? 1.f : -1.f) * M->heli_half, (k & 4 ? 1.f : -1.f) * M->heli_half};
        V3 xc = mul(R, loc);
        float d = -(B.pos.z + xc.z);
        if (d <= -kContactMargin) continue;
        V3 vp = v0.l + cross(v0.a, xc);
        float w = ramp01(fmaxf(d, d - h * vp.z), M->pen_ramp);
        float gn = w * (h * M->heli_gnd_k + M->heli_gnd_c);
        float fn = fmaxf(w * M->heli_gnd_k * d - gn * vp.z, 0.f);
        if (!(w > 0.f)) continue;
        Contact ct = contact_none();
        ct.active = 1.f; ct.xc = xc; ct.n = V3{0, 0, 1}; ct.kd = w * M->heli_gnd_k * d; ct.gn = gn;
        float vt = sqrtf(vp.x * vp.x + vp.y * vp.y);
        ct.ct = M->gnd_mu * fn / fmaxf(vt, M->slip_eps);
        ct.vrel = vp;
        contact_fold(ct, h, A, p);
    }
    S6 a = solve6(A, S6{V3{-p.a.x, -p.a.y, -p.a.z}, V3{-p.l.x, -p.l.y, -p.l.z}});
    V3 wxv = cross(v0.a, v0.l);
    B.vel = B.vel + h * (a.l + wxv);
    B.ang = B.ang + h * a.a;
    clamp_angvel(B.ang, M->heli_max_angvel);
    B.pos = B.pos + h * B.vel;
    quat_integrate(B.qx, B.qy, B.qz, B.qw, B.ang, h);
}

// ---------------------------------------------------------------------------------------------
// observation / reward helpers (oracle: ant_obs_core_compute etc.; reference citations there)
// ---------------------------------------------------------------------------------------------
MMS_HD V3 quat_rotate(float x, float y, float z, float w, V3 v, float sign) {
    V3 qv = V3{x, y, z};
    float a = 2.0f * w * w - 1.0f;
    V3 c = cross(qv, v);
    float d = dot(qv, v);
    return V3{v.x * a + sign * (c.x * w * 2.0f) + x * d * 2.0f, v.y * a + sign * (c.y * w * 2.0f) + y * d * 2.0f,
              v.z * a + sign * (c.z * w * 2.0f) + z * d * 2.0f};
}
// a % (2 pi) with the sign of the divisor (torch's %), for |a| < 2 pi -- every caller passes an atan2f result, for which
// fmodf(a, 2 pi) returns a itself: the generic fmodf costs ~60 instructions on the GPU for nothing
MMS_HD float wrap_2pi(float a) { return (a < 0.f) ? a + kTwoPi : a; }
struct AntObsCore { V3 vel_loc, angvel_loc; float yaw, roll, angle_to_target, up_proj, heading_proj; };
// p: GLOBAL position (local + env origin)
MMS_HD AntObsCore ant_obs_core(V3 p, float x, float y, float z, float w, V3 vel, V3 ang) {
    AntObsCore o;
    V3 to_target = V3{0.f - p.x, 0.f - p.y, 0.0f};
    float n = sqrtf(dot(to_target, to_target));
    if (n < 1e-9f) n = 1e-9f;
    V3 td = V3{to_target.x / n, to_target.y / n, to_target.z / n};
    // torso_quat = quat_mul(q, conj(identity)) with the -0.0 terms of the reference kept
    float tx = w * (-0.f) + x * 1.f + y * (-0.f) - z * (-0.f);
    float ty = w * (-0.f) - x * (-0.f) + y * 1.f + z * (-0.f);
    float tz = w * (-0.f) + x * (-0.f) - y * (-0.f) + z * 1.f;
    float tw = w * 1.f - x * (-0.f) - y * (-0.f) - z * (-0.f);
    V3 up = quat_rotate(tx, ty, tz, tw, V3{0, 0, 1}, 1.f);
    V3 hd = quat_rotate(tx, ty, tz, tw, V3{1, 0, 0}, 1.f);
    o.up_proj = up.z;
    o.heading_proj = hd.x * td.x + hd.y * td.y + hd.z * td.z;
    o.vel_loc = quat_rotate(tx, ty, tz, tw, vel, -1.f);
    o.angvel_loc = quat_rotate(tx, ty, tz, tw, ang, -1.f);
    float sinr = 2.0f * (tw * tx + ty * tz), cosr = tw * tw - tx * tx - ty * ty + tz * tz;
    o.roll = wrap_2pi(atan2f(sinr, cosr));
    float siny = 2.0f * (tw * tz + tx * ty), cosy = tw * tw + tx * tx - ty * ty - tz * tz;
    o.yaw = wrap_2pi(atan2f(siny, cosy));
    float walk = atan2f(0.f - p.z, 0.f - p.x);
    o.angle_to_target = walk - o.yaw;
    return o;
}
MMS_HD float unscale1(float x, float lo, float hi) { return (2.0f * x - hi - lo) / (hi - lo); }
MMS_HD float l2_dist2(float ax, float ay, float bx, float by) {
    float c1 = ax - bx, c2 = ay - by;
    return sqrtf(c1 * c1 + c2 * c2);
}
MMS_HD float box_angle(float qz, float qw) { return atanf((2.f * qw * qz) / (1.f - 2.f * qz * qz)); }
// (sin, -cos) of that angle (ten_ant.py:1353-1393 offsets the goals along it) without the trigonometry: the angle is
// atan(t) in (-pi/2, pi/2), so sin = t / sqrt(1 + t^2) and cos = 1 / sqrt(1 + t^2); written on (num, den) so that
// den = 0 (a quarter turn) gives (+-1, 0) like sin / cos of atan(+-inf)
MMS_HD void box_yaw_dir(float qz, float qw, float& sv, float& cv) {
    float num = 2.f * qw * qz, den = 1.f - 2.f * qz * qz;
    float inv = 1.f / sqrtf(num * num + den * den);
    float sgn = (den < 0.f) ? -1.f : 1.f;                  // atan(num / den): the sign of den folds into the quotient
    sv = sgn * num * inv;
    cv = -(sgn * den * inv);
}
MMS_HD float box_quat_dist(float qx, float qy, float qz, float qw) {
    float x = 2.f * (qx * qy + qw * qz), y = 1.f - 2.f * (qx * qx + qz * qz), z = 2.f * (qy * qz - qw * qx);
    float x1 = x * 0.f, y1 = y * 1.f, z1 = z * 0.f;
    return (x1 + y1 + z1) / sqrtf(x * x + y * y + z * z) / sqrtf(0.f * 0.f + 1.f * 1.f + 0.f * 0.f);
}

// counter-based RNG (oracle mo_rand_uniform)
MMS_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
MMS_HD float rand_uniform(uint64_t seed, uint64_t env_global, uint64_t step, uint32_t k) {
    uint32_t x = mix32((uint32_t)seed ^ 0x9E3779B9U);
    x = mix32(x ^ (uint32_t)(seed >> 32));
    x = mix32(x ^ (uint32_t)env_global);
    x = mix32(x ^ (uint32_t)(env_global >> 32) ^ 0x85EBCA6BU);
    x = mix32(x ^ (uint32_t)step);
    x = mix32(x ^ (uint32_t)(step >> 32) ^ (k * 0xC2B2AE35U));
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// standard normal by Box-Muller from two counter-based uniforms (oracle: mo_rand_normal); the first is moved off zero
MMS_HD float rand_normal(uint64_t seed, uint64_t row, uint64_t counter, uint32_t k) {
    float u1 = rand_uniform(seed, row, counter, 2u * k) + (0.5f / 16777216.0f);
    float u2 = rand_uniform(seed, row, counter, 2u * k + 1u);
    return sqrtf(-2.0f * logf(u1)) * cosf(kTwoPi * u2);
}

// ---------------------------------------------------------------------------------------------
// epilogue of the ant tasks: reset, observation row, reward partials (per lane)
// ---------------------------------------------------------------------------------------------
// reward partial slots per ant in the reduction scratch
enum { RP_ADR = 0, RP_GDR, RP_GAR, RP_UP, RP_EC, RP_LIM, RP_FALLEN, RP_ACOST, RP_STRIDE };

// reset_idx for one leg lane (ten_ant.py:810-868): root <- initial, dof <- clamp(init + noise), vel <- noise
MMS_HD void ant_reset_lane(const mms_config* C, const LegConst& L, AntLane& S, const float* init_root13, int leg,
                           const float* ext_noise16, uint64_t env_global, uint64_t reset_count) {
    S.pos = V3{init_root13[0], init_root13[1], init_root13[2]};
    S.qx = init_root13[3]; S.qy = init_root13[4]; S.qz = init_root13[5]; S.qw = init_root13[6];
    S.vel = V3{init_root13[7], init_root13[8], init_root13[9]};
    S.ang = V3{init_root13[10], init_root13[11], init_root13[12]};
    for (int j = 0; j < 2; j++) {
        int d = 2 * leg + j;
        float npos, nvel;
        if (C->external_noise) { npos = ext_noise16[d]; nvel = ext_noise16[8 + d]; }
        else {
            npos = 0.4f * rand_uniform(C->seed, env_global, reset_count, (uint32_t)d) - 0.2f;
            nvel = 0.2f * rand_uniform(C->seed, env_global, reset_count, (uint32_t)(8 + d)) - 0.1f;
        }
        S.q[j] = clampf(L.init[j] + npos, L.lower[j], L.upper[j]);
        S.qd[j] = nvel;
    }
}

// goal of ant k (ten_ant.py:1353-1393): box + / - (1.5 + 3 (k/2)) (sin, -cos)(box yaw)
MMS_HD void tenant_goal(int k, float bx, float by, float sv, float cv, float& gx, float& gy) {
    float off = 1.5f + 3.0f * (float)(k / 2);
    if (k % 2 == 0) { gx = bx + off * sv; gy = by + off * cv; }
    else { gx = bx - off * sv; gy = by - off * cv; }
}

// TenAnt: observation pieces of one leg lane into the staged row, and the per-lane reward partials.
// ec / lim / acost are this lane's share; the caller quad-reduces them and lane 0 stores the ant's slots.
struct TenAntLaneOut { float adr, gdr, gar, up, fallen, ec, lim, acost; float px, py, gx, gy; };
MMS_HD TenAntLaneOut tenant_obs_reward_lane(const mms_config* C, const LegConst& L, const AntLane& S, int ant, int leg,
                                            V3 origin, float act0, float act1, float box_gx, float box_gy, float sv, float cv,
                                            float pbx, float pby, float gbx, float gby, float* s_obs) {
    TenAntLaneOut o;
    V3 p = S.pos + origin;                                   // global frame
    AntObsCore c = ant_obs_core(p, S.qx, S.qy, S.qz, S.qw, S.vel, S.ang);
    float* row = s_obs + 38 * ant;
    if (leg == 0) {
        row[0] = p.x; row[1] = p.y; row[2] = p.z;
        row[3] = c.vel_loc.x; row[4] = c.vel_loc.y; row[5] = c.vel_loc.z;
        row[6] = c.angvel_loc.x; row[7] = c.angvel_loc.y; row[8] = c.angvel_loc.z;
        row[9] = c.yaw; row[10] = c.roll; row[11] = c.angle_to_target; row[12] = c.up_proj; row[13] = c.heading_proj;
    }
    float act[2] = {act0, act1};
    o.ec = 0.f; o.lim = 0.f; o.acost = 0.f;
    for (int j = 0; j < 2; j++) {
        float us = unscale1(S.q[j], L.lower[j], L.upper[j]);
        float dv = S.qd[j] * C->dof_vel_scale;
        row[14 + 2 * leg + j] = us;
        row[22 + 2 * leg + j] = dv;
        row[30 + 2 * leg + j] = act[j];
        o.ec += fabsf(act[j] * dv);
        o.lim += (us > 0.99f) ? 1.f : 0.f;
        o.acost += act[j] * act[j];
    }
    float gx, gy;
    tenant_goal(ant, box_gx, box_gy, sv, cv, gx, gy);
    float off = 1.5f + 3.0f * (float)(ant / 2);
    float btx = 0.f, bty = (ant % 2 == 0) ? -off : off;     // box_targets_k (ten_ant.py:172-181)
    float d_now = l2_dist2(p.x, p.y, gx, gy);
    float push = (d_now < 1.5f) ? 0.f : 1.f;
    float ant_dist = l2_dist2(pbx, pby, gbx, gby) - d_now;
    o.adr = C->ant_dist_reward_scale * ant_dist * push;
    float gd_before = l2_dist2(btx, bty, gbx, gby);
    float gd = l2_dist2(btx, bty, gx, gy);
    o.gdr = C->goal_dist_reward_scale * (gd_before - gd);
    o.gar = (gd < 0.5f) ? 1.f : 0.f;                         // arrive flag; reward adds 2 per arrival
    o.up = (c.up_proj > 0.93f) ? (0.f + C->up_weight) : 0.f;
    o.fallen = (p.z < C->termination_height) ? 1.f : 0.f;
    o.px = p.x; o.py = p.y; o.gx = gx; o.gy = gy;
    return o;
}

// TenAnt: final reduction over the ants in reference order (ten_ant.py:1173-1301); one thread.
MMS_HD void tenant_reward_finish(const mms_config* C, int A, const float* s_red, float bqx, float bqy, float bqz, float bqw,
                                 int64_t progress, float& rew, int64_t& reset) {
    float quat_dist = box_quat_dist(bqx, bqy, bqz, bqw);
    float quat_reward = C->quat_reward_scale * quat_dist;
    float adr = 0.f, gdr = 0.f, gar = 0.f, up = 0.f, ec = 0.f, acost = 0.f;
    float lim = 0.f;
    bool all_arrive = true, fallen = false;
    for (int k = 0; k < A; k++) {
        const float* r = s_red + RP_STRIDE * k;
        float g2 = (r[RP_GAR] != 0.f) ? 2.f : 0.f;
        adr = (k == 0) ? r[RP_ADR] : adr + r[RP_ADR];
        gdr = (k == 0) ? r[RP_GDR] : gdr + r[RP_GDR];
        gar = (k == 0) ? g2 : gar + g2;
        up = (k == 0) ? r[RP_UP] : up + r[RP_UP];
        ec = (k == 0) ? r[RP_EC] : ec + r[RP_EC];
        lim += r[RP_LIM];
        acost += r[RP_ACOST];
        all_arrive = all_arrive && (r[RP_GAR] != 0.f);
        fallen = fallen || (r[RP_FALLEN] != 0.f);
    }
    up = up * 10.f;
    float success = ((quat_dist > 0.9f) && all_arrive) ? 100.f : 0.f;
    float total = 5.f + up + quat_reward + adr + gdr + gar + success - C->actions_cost * acost - C->energy_cost * ec -
                  lim * C->joints_at_limit_cost;
    if (fallen) total = C->death_cost;
    int64_t rs = fallen ? 1 : 0;                              // reset_buf is 0 here: reset_idx cleared it
    if (progress >= (int64_t)C->max_episode_length - 1) rs = 1;
    rew = total;
    reset = rs;
}

// MultiAntCircle (multi_ant_circle.py:385-543; INTENDED semantics, see include/mms.h): the 38 observation entries per ant are
// TenAnt's (target = origin); the per-ant reward pieces of one leg lane.  rk rides in the RP_ADR slot of the reduction scratch.
MMS_HD float circle_angle(float a, float b) {                  // compute_angle (:385-398): degrees in [0, 360)
    float deg = fabsf(atan2f(b, a) * 180.f / 3.141592653589793f);
    return (b < 0.f) ? 360.f + (-1.f) * deg : deg;
}
struct CircleLaneOut { float rk, up, fallen, ec, lim, acost; float px, py; };
MMS_HD CircleLaneOut circle_obs_reward_lane(const mms_config* C, const LegConst& L, const AntLane& S, int ant, int leg, V3 origin,
                                            float act0, float act1, float pbx, float pby, float* s_obs) {
    CircleLaneOut o;
    V3 p = S.pos + origin;                                   // global frame
    AntObsCore c = ant_obs_core(p, S.qx, S.qy, S.qz, S.qw, S.vel, S.ang);
    float* row = s_obs + 38 * ant;
    if (leg == 0) {
        row[0] = p.x; row[1] = p.y; row[2] = p.z;
        row[3] = c.vel_loc.x; row[4] = c.vel_loc.y; row[5] = c.vel_loc.z;
        row[6] = c.angvel_loc.x; row[7] = c.angvel_loc.y; row[8] = c.angvel_loc.z;
        row[9] = c.yaw; row[10] = c.roll; row[11] = c.angle_to_target; row[12] = c.up_proj; row[13] = c.heading_proj;
    }
    float act[2] = {act0, act1};
    o.ec = 0.f; o.lim = 0.f; o.acost = 0.f;
    for (int j = 0; j < 2; j++) {
        float us = unscale1(S.q[j], L.lower[j], L.upper[j]);
        float dv = S.qd[j] * C->dof_vel_scale;
        row[14 + 2 * leg + j] = us;
        row[22 + 2 * leg + j] = dv;
        row[30 + 2 * leg + j] = act[j];
        o.ec += fabsf(act[j] * dv);
        o.lim += (us > 0.99f) ? 1.f : 0.f;
        o.acost += act[j] * act[j];
    }
    float sgn = (ant % 2 == 0) ? 1.f : -1.f;                 // pos_2 = -obs_buf_2[:, :2] (:428); its cached position is not negated (:431)
    float px = sgn * p.x, py = sgn * p.y;
    float dist = sqrtf(px * px + py * py);
    float ang = circle_angle(px, py), ang_before = circle_angle(pbx, pby);
    bool on = (ang - ang_before > 0.f) && (dist >= 2.7f) && (dist <= 3.3f);
    o.rk = (on ? 2.f : 0.f) + ((on ? 1.f : 0.f) - 1.f);
    o.up = (c.up_proj > 0.93f) ? (0.f + C->up_weight) : 0.f;
    o.fallen = (p.z < C->termination_height) ? 1.f : 0.f;
    o.px = p.x; o.py = p.y;
    return o;
}
MMS_HD void circle_reward_finish(const mms_config* C, int A, const float* s_red, int64_t progress, float& rew, int64_t& reset) {
    float rk = 0.f, up = 0.f, ec = 0.f, acost = 0.f, lim = 0.f;
    bool fallen = false;
    for (int k = 0; k < A; k++) {
        const float* r = s_red + RP_STRIDE * k;
        rk = (k == 0) ? r[RP_ADR] : rk + r[RP_ADR];
        up = (k == 0) ? r[RP_UP] : up + r[RP_UP];
        ec = (k == 0) ? r[RP_EC] : ec + r[RP_EC];
        acost = (k == 0) ? r[RP_ACOST] : acost + r[RP_ACOST];
        lim += r[RP_LIM];
        fallen = fallen || (r[RP_FALLEN] != 0.f);
    }
    float total = up + rk - C->actions_cost * acost - C->energy_cost * ec - lim * C->joints_at_limit_cost;
    if (fallen) total = C->death_cost;
    int64_t rs = fallen ? 1 : 0;
    if (progress >= (int64_t)C->max_episode_length - 1) rs = 1;
    rew = total;
    reset = rs;
}

// OneAnt (one_ant.py:465-627): the four leg lanes fill the 60-wide row; lane 0 computes the reward.
struct OneAntLaneOut { float ec, lim, acost; };
MMS_HD OneAntLaneOut oneant_obs_lane(const mms_config* C, const LegConst& L, const AntLane& S, int leg, V3 origin,
                                     float act0, float act1, const float sens[6], float* s_obs, AntObsCore& core, V3& pglob) {
    OneAntLaneOut o;
    V3 p = S.pos + origin;
    pglob = p;
    core = ant_obs_core(p, S.qx, S.qy, S.qz, S.qw, S.vel, S.ang);
    if (leg == 0) {
        s_obs[0] = p.z;
        s_obs[1] = core.vel_loc.x; s_obs[2] = core.vel_loc.y; s_obs[3] = core.vel_loc.z;
        s_obs[4] = core.angvel_loc.x; s_obs[5] = core.angvel_loc.y; s_obs[6] = core.angvel_loc.z;
        s_obs[7] = core.yaw; s_obs[8] = core.roll; s_obs[9] = core.angle_to_target; s_obs[10] = core.up_proj;
        s_obs[11] = core.heading_proj;
    }
    float act[2] = {act0, act1};
    o.ec = 0.f; o.lim = 0.f; o.acost = 0.f;
    for (int j = 0; j < 2; j++) {
        float us = unscale1(S.q[j], L.lower[j], L.upper[j]);
        float dv = S.qd[j] * C->dof_vel_scale;
        s_obs[12 + 2 * leg + j] = us;
        s_obs[20 + 2 * leg + j] = dv;
        s_obs[52 + 2 * leg + j] = act[j];
        o.ec += fabsf(act[j] * dv);
        o.lim += (us > 0.99f) ? 1.f : 0.f;
        o.acost += act[j] * act[j];
    }
    for (int i = 0; i < 6; i++) s_obs[28 + 6 * leg + i] = sens[i] * C->contact_force_scale;
    return o;
}
MMS_HD void oneant_reward(const mms_config* C, float obs0, float up_proj, float ec, float lim, float acost, float pbx,
                          float pby, float bbx, float bby, float ax, float ay, float bx, float by, float bqx, float bqy,
                          float bqz, float bqw, int64_t progress, float& rew, int64_t& reset) {
    float quat_dist = box_quat_dist(bqx, bqy, bqz, bqw);
    float quat_reward = C->quat_reward_scale * quat_dist;
    float d_now = l2_dist2(ax, ay, bx, by);
    float push = (d_now < 1.5f) ? 0.f : 1.f;
    float ant_dist = l2_dist2(pbx, pby, bbx, bby) - d_now;
    float adr = C->ant_dist_reward_scale * ant_dist * push;
    float gd_before = l2_dist2(0.f, 0.f, bbx, bby);
    float gd = l2_dist2(0.f, 0.f, bx, by);
    bool arrive = gd < 0.5f;
    float gdr = C->goal_dist_reward_scale * (gd_before - gd);
    float gar = arrive ? 2.f : 0.f;
    float success = ((quat_dist > 0.9f) && arrive) ? 10.f : 0.f;
    float up = (up_proj > 0.93f) ? (0.f + C->up_weight) : 0.f;
    float total = 0.5f + up + quat_reward + adr + gdr + gar + success - C->actions_cost * acost - C->energy_cost * ec -
                  lim * C->joints_at_limit_cost;
    bool fallen = obs0 < C->termination_height;
    if (fallen) total = C->death_cost;
    int64_t rs = fallen ? 1 : 0;
    if (progress >= (int64_t)C->max_episode_length - 1) rs = 1;
    rew = total;
    reset = rs;
}

// MultiIngenuity reward (multi_ingenuity.py:381-453) from the global-frame observation row [4][13]
MMS_HD void ingenuity_reward(const float* roots, int32_t max_len, int64_t progress, float& rew, int64_t& reset) {
    const float gy[4] = {2.f, -2.f, 6.f, -6.f};
    float pos_reward = 0.f, up_reward = 0.f, spin_reward = 0.f;
    bool die = false, low = false;
    for (int k = 0; k < 4; k++) {
        const float* r = roots + 13 * k;
        float dx = 4.f - r[0], dy = gy[k] - r[1], dz = 1.f - r[2];
        float td = sqrtf(dx * dx + dy * dy + dz * dz);
        float pr = 1.0f / (1.0f + td * td);
        pos_reward = (k == 0) ? pr : pos_reward + pr;
        V3 ups = quat_rotate(r[3], r[4], r[5], r[6], V3{0, 0, 1}, 1.f);
        float tilt = fabsf(1.f - ups.z);
        float ur = 5.0f / (1.0f + tilt * tilt);
        up_reward = (k == 0) ? ur : up_reward + ur;
        float spin = fabsf(r[12]);
        float sr = 1.0f / (1.0f + spin * spin);
        spin_reward = (k == 0) ? sr : spin_reward + sr;
        die = die || (td > 8.0f);
        low = low || (r[2] < 0.5f);
    }
    rew = pos_reward + pos_reward * (up_reward + spin_reward);
    int64_t d = (die || low) ? 1 : 0;
    reset = (progress >= (int64_t)max_len - 1) ? 1 : d;
}

}  // namespace mms
