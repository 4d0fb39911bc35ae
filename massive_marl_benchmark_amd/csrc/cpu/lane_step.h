// lane_step.h -- one environment's VecTask step on the HOST, built from the SAME per-lane functions as the HIP kernels
// (../mms_lane.h): the lanes of an env run as plain loops, the DPP / LDS reductions become explicit sums in the same association
// order.  Two users:
//   * cpu/mms_cpu.cpp  -> lib/libmms_cpu.so, the explicit opt-in CPU build of the engine behind the C ABI (mms_config.device = -1;
//                         BASELINE configs[0] "OneAnt num_envs=64, sim_device=cpu", reference base_task.py:27-32);
//   * tests/emu/emu_step.cpp, the lane-emulation test harness (tests/test_lane_emulation.py).
// This is product code, not the oracle: it shares no source with oracle/mms_oracle.c.
#pragma once
#include <string.h>

#include <vector>

#include "../mms_lane.h"

namespace mms {

static RigidState load_rigid(const float* r) {
    RigidState B;
    B.pos = V3{r[0], r[1], r[2]};
    B.qx = r[3]; B.qy = r[4]; B.qz = r[5]; B.qw = r[6];
    B.vel = V3{r[7], r[8], r[9]};
    B.ang = V3{r[10], r[11], r[12]};
    return B;
}
static void store_rigid(float* r, const RigidState& B) {
    r[0] = B.pos.x; r[1] = B.pos.y; r[2] = B.pos.z; r[3] = B.qx; r[4] = B.qy; r[5] = B.qz; r[6] = B.qw;
    r[7] = B.vel.x; r[8] = B.vel.y; r[9] = B.vel.z; r[10] = B.ang.x; r[11] = B.ang.y; r[12] = B.ang.z;
}
static float quad4(const float x[4]) { return (x[0] + x[1]) + (x[2] + x[3]); }

// fp32 <-> fp16 as the device converts (v_cvt_f16_f32: round to nearest even, subnormals kept)
static inline uint16_t f2h(float f) {                                     // round to nearest even, subnormals kept, as v_cvt_f16_f32
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);               // rounds to 65520 or more: infinity
    if (u < 0x38800000u) {                                                // below 2^-14: a subnormal half (or zero)
        if (u < 0x33000000u) return (uint16_t)sign;                       // below 2^-25: zero
        const int e = (int)(u >> 23);                                     // biased exponent, 102 .. 112
        uint32_t m = (u & 0x7fffffu) | 0x800000u;                         // 24-bit significand
        const int shift = 126 - e;                                        // the half's unit is 2^-24: value = m 2^(e - 150) = (m >> shift) 2^-24
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t r = u + 0xfffu + ((u >> 13) & 1u);                      // round the 13 dropped bits to nearest even
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
static inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {
            const float v = (float)m * 5.9604644775390625e-08f;           // m 2^-24, exact
            memcpy(&u, &v, 4);
            u |= sign;
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// The engine's buffers (include/mms.h: mms_get_tensor names).  obs / obs_clipped / obs_out / rew_out / done_out may be NULL
// (mms_set_obs_outputs, mms_bind_obs_out, mms_bind_rollout_out); dr is NULL unless mms_set_dr(h, 1).
struct HostBufs {
    const float* actions; float* obs; float* obs_clipped; float* rew; int64_t* reset; int64_t* progress;
    float* root_states; const float* initial_root_states; float* dof_state; const float* env_origin; float* prev;
    const float* reset_noise; float* foot_sensors; int64_t* reset_count; const float* dr;
    float* obs_out = nullptr; float* rew_out = nullptr; uint8_t* done_out = nullptr;
    uint16_t* obs_planes = nullptr; float obs_planes_scale = 1.f;      // mms_bind_obs_planes16 (H32 planes of the clamped row)
};

static void host_ant_env(const mms_config* C, const HostBufs& b, int env, int do_physics, int obs_dim, int prev_dim) {
    const mms_model* M = &C->model;
    const int A = C->num_agents, task = C->task, nl = 4 * A, actors = A + 1;
    float* root_env = b.root_states + (size_t)env * actors * 13;
    const float* init_env = b.initial_root_states + (size_t)env * actors * 13;
    float* dof_env = b.dof_state + (size_t)env * A * 16;
    const float* act_env = b.actions + (size_t)env * A * 8;
    float* prev_env = b.prev + (size_t)env * prev_dim;
    V3 origin = V3{b.env_origin[3 * env], b.env_origin[3 * env + 1], b.env_origin[3 * env + 2]};
    int64_t reset_flag = b.reset[env], progress = b.progress[env];
    uint64_t env_global = (uint64_t)(C->env_offset + env);
    std::vector<LegConst> L(nl);
    std::vector<AntLane> S(nl);
    std::vector<float> act0(nl), act1(nl);
    std::vector<float> sens(nl * 6, 0.f);
    for (int t = 0; t < nl; t++) {
        int ant = t >> 2, leg = t & 3;
        L[t] = load_leg_const(M, leg);
        const float* r = root_env + 13 * ant;
        S[t].pos = V3{r[0], r[1], r[2]}; S[t].qx = r[3]; S[t].qy = r[4]; S[t].qz = r[5]; S[t].qw = r[6];
        S[t].vel = V3{r[7], r[8], r[9]}; S[t].ang = V3{r[10], r[11], r[12]};
        S[t].q[0] = dof_env[4 * t]; S[t].qd[0] = dof_env[4 * t + 1]; S[t].q[1] = dof_env[4 * t + 2]; S[t].qd[1] = dof_env[4 * t + 3];
        act0[t] = clampf(act_env[2 * t], -C->clip_actions, C->clip_actions);
        act1[t] = clampf(act_env[2 * t + 1], -C->clip_actions, C->clip_actions);
        if (task == MMS_TASK_ONE_ANT)
            for (int i = 0; i < 6; i++) sens[6 * t + i] = b.foot_sensors[((size_t)env * A + ant) * 24 + 6 * leg + i];
    }
    RigidState B = load_rigid(root_env + 13 * A);
    if (do_physics && reset_flag == 0) {
        float h = C->dt / (float)C->substeps;
        for (int s = 0; s < C->substeps; s++) {
            BoxPose bp;
            bp.pos = B.pos; bp.R = quat_to_mat(B.qx, B.qy, B.qz, B.qw); bp.v = B.vel; bp.w = B.ang;
            bp.half = V3{M->box_half[0], M->box_half[1], M->box_half[2]};
            std::vector<Sym6> IA(nl);
            std::vector<S6> pA(nl), wr(nl);
            std::vector<LegPass> P(nl);
            std::vector<SensorPass> SP(nl);
            const KinPark no_park{nullptr, 0};
            for (int t = 0; t < nl; t++) {
                float t1 = act0[t] * L[t].gear[0] * C->power_scale, t2 = act1[t] * L[t].gear[1] * C->power_scale;
                if (b.dr) {
                    LegDR d = load_leg_dr(b.dr + ((size_t)env * A + (t >> 2)) * MMS_DR_FLOATS, t & 3);
                    if (task == MMS_TASK_ONE_ANT) leg_inward<true, true>(M, L[t], h, S[t], t & 3, t1, t2, true, bp, P[t], &SP[t], IA[t], pA[t], no_park, &d);
                    else leg_inward<false, true>(M, L[t], h, S[t], t & 3, t1, t2, true, bp, P[t], &SP[t], IA[t], pA[t], no_park, &d);
                } else if (task == MMS_TASK_ONE_ANT) leg_inward<true>(M, L[t], h, S[t], t & 3, t1, t2, true, bp, P[t], &SP[t], IA[t], pA[t]);
                else leg_inward<false>(M, L[t], h, S[t], t & 3, t1, t2, true, bp, P[t], &SP[t], IA[t], pA[t]);
            }
            for (int q = 0; q < nl; q += 4) {                      // quad all-reduce
                Sym6 sum;
                S6 ps;
                for (int k = 0; k < 21; k++) { float x[4] = {IA[q].m[k], IA[q + 1].m[k], IA[q + 2].m[k], IA[q + 3].m[k]}; sum.m[k] = quad4(x); }
                float* pp[6] = {&ps.a.x, &ps.a.y, &ps.a.z, &ps.l.x, &ps.l.y, &ps.l.z};
                for (int k = 0; k < 6; k++) { float x[4] = {get(pA[q], k), get(pA[q + 1], k), get(pA[q + 2], k), get(pA[q + 3], k)}; *pp[k] = quad4(x); }
                for (int j = 0; j < 4; j++) { IA[q + j] = sum; pA[q + j] = ps; }
            }
            for (int t = 0; t < nl; t++) {
                if (task == MMS_TASK_ONE_ANT) leg_outward<true>(M, L[t], h, S[t], t & 3, bp, P[t], &SP[t], IA[t], pA[t], wr[t], &sens[6 * t]);
                else leg_outward<false>(M, L[t], h, S[t], t & 3, bp, P[t], &SP[t], IA[t], pA[t], wr[t], nullptr);
            }
            float wt[6];
            for (int c = 0; c < 6; c++) { float tsum = 0.f; for (int t = 0; t < nl; t++) tsum += get(wr[t], c); wt[c] = tsum; }
            if (M->boxgnd_mu > 0.f) {
                BoxCornerF bc[8], bt;
                for (int c = 0; c < 8; c++) bc[c] = box_corner_friction(M, h, B, bp.R, c);
                auto osum = [&](auto get) { return ((get(0) + get(1)) + (get(2) + get(3))) + ((get(4) + get(5)) + (get(6) + get(7))); };
                for (int k = 0; k < 21; k++) bt.IA.m[k] = osum([&](int c) { return bc[c].IA.m[k]; });
                float* pp[6] = {&bt.pA.a.x, &bt.pA.a.y, &bt.pA.a.z, &bt.pA.l.x, &bt.pA.l.y, &bt.pA.l.z};
                for (int k = 0; k < 6; k++) *pp[k] = osum([&](int c) { return get(bc[c].pA, k); });
                box_finish_friction(M, h, B, bp.R, bt, S6{V3{wt[0], wt[1], wt[2]}, V3{wt[3], wt[4], wt[5]}});
            } else {
            BoxCorner bc[8], bt;
            for (int c = 0; c < 8; c++) bc[c] = box_corner(M, h, B, bp.R, c);
            for (int k = 0; k < 9; k++) {                          // quad sums, then the half-mirror pair
                float q0 = (bc[0].t[k] + bc[1].t[k]) + (bc[2].t[k] + bc[3].t[k]), q1 = (bc[4].t[k] + bc[5].t[k]) + (bc[6].t[k] + bc[7].t[k]);
                bt.t[k] = q0 + q1;
            }
            box_finish(M, h, B, bp.R, bt, S6{V3{wt[0], wt[1], wt[2]}, V3{wt[3], wt[4], wt[5]}});
            }
        }
    }
    progress += 1;
    if (reset_flag != 0) {
        for (int t = 0; t < nl; t++) ant_reset_lane(C, L[t], S[t], init_env + 13 * (t >> 2), t & 3, b.reset_noise + 16 * (size_t)env, env_global, (uint64_t)b.reset_count[env]);
        B = load_rigid(init_env + 13 * A);
        progress = 0;
        b.reset_count[env] += 1;
    }
    for (int t = 0; t < nl; t++) {
        dof_env[4 * t] = S[t].q[0]; dof_env[4 * t + 1] = S[t].qd[0]; dof_env[4 * t + 2] = S[t].q[1]; dof_env[4 * t + 3] = S[t].qd[1];
        if ((t & 3) == 0) {
            RigidState R;
            R.pos = S[t].pos; R.qx = S[t].qx; R.qy = S[t].qy; R.qz = S[t].qz; R.qw = S[t].qw; R.vel = S[t].vel; R.ang = S[t].ang;
            store_rigid(root_env + 13 * (t >> 2), R);
        }
    }
    store_rigid(root_env + 13 * A, B);
    std::vector<float> s_obs(obs_dim, 0.f), s_red(RP_STRIDE * A, 0.f);
    float bgx = B.pos.x + origin.x, bgy = B.pos.y + origin.y;
    float rew;
    int64_t rs;
    if (task == MMS_TASK_TEN_ANT) {
        float sv, cv;
        box_yaw_dir(B.qz, B.qw, sv, cv);
        std::vector<TenAntLaneOut> o(nl);
        std::vector<float> newprev(prev_dim);
        for (int t = 0; t < nl; t++) {
            int ant = t >> 2;
            float pbx = prev_env[2 * ant], pby = prev_env[2 * ant + 1], gbx = prev_env[2 * A + 2 * ant], gby = prev_env[2 * A + 2 * ant + 1];
            o[t] = tenant_obs_reward_lane(C, L[t], S[t], ant, t & 3, origin, act0[t], act1[t], bgx, bgy, sv, cv, pbx, pby, gbx, gby, s_obs.data());
        }
        for (int q = 0; q < nl; q += 4) {
            float e[4] = {o[q].ec, o[q + 1].ec, o[q + 2].ec, o[q + 3].ec}, l[4] = {o[q].lim, o[q + 1].lim, o[q + 2].lim, o[q + 3].lim},
                  c[4] = {o[q].acost, o[q + 1].acost, o[q + 2].acost, o[q + 3].acost};
            float* r = s_red.data() + RP_STRIDE * (q >> 2);
            r[RP_ADR] = o[q].adr; r[RP_GDR] = o[q].gdr; r[RP_GAR] = o[q].gar; r[RP_UP] = o[q].up; r[RP_EC] = quad4(e); r[RP_LIM] = quad4(l);
            r[RP_FALLEN] = o[q].fallen; r[RP_ACOST] = quad4(c);
            int ant = q >> 2;
            prev_env[2 * ant] = o[q].px; prev_env[2 * ant + 1] = o[q].py; prev_env[2 * A + 2 * ant] = o[q].gx; prev_env[2 * A + 2 * ant + 1] = o[q].gy;
        }
        float* tt = s_obs.data() + 38 * A;
        tt[0] = bgx; tt[1] = bgy; tt[2] = B.qx; tt[3] = B.qy; tt[4] = B.qz; tt[5] = B.qw; tt[6] = 0.f; tt[7] = 0.f;
        prev_env[4 * A] = bgx; prev_env[4 * A + 1] = bgy;
        tenant_reward_finish(C, A, s_red.data(), B.qx, B.qy, B.qz, B.qw, progress, rew, rs);
    } else if (task == MMS_TASK_MULTI_ANT_CIRCLE) {
        std::vector<CircleLaneOut> o(nl);
        for (int t = 0; t < nl; t++) {
            int ant = t >> 2;
            o[t] = circle_obs_reward_lane(C, L[t], S[t], ant, t & 3, origin, act0[t], act1[t], prev_env[2 * ant], prev_env[2 * ant + 1], s_obs.data());
        }
        for (int q = 0; q < nl; q += 4) {
            float e[4] = {o[q].ec, o[q + 1].ec, o[q + 2].ec, o[q + 3].ec}, l[4] = {o[q].lim, o[q + 1].lim, o[q + 2].lim, o[q + 3].lim},
                  c[4] = {o[q].acost, o[q + 1].acost, o[q + 2].acost, o[q + 3].acost};
            float* r = s_red.data() + RP_STRIDE * (q >> 2);
            r[RP_ADR] = o[q].rk; r[RP_UP] = o[q].up; r[RP_EC] = quad4(e); r[RP_LIM] = quad4(l); r[RP_FALLEN] = o[q].fallen; r[RP_ACOST] = quad4(c);
            prev_env[2 * (q >> 2)] = o[q].px; prev_env[2 * (q >> 2) + 1] = o[q].py;
        }
        circle_reward_finish(C, A, s_red.data(), progress, rew, rs);
    } else {
        float pot_in = prev_env[4];
        OneAntLaneOut o[4];
        AntObsCore core;
        V3 pg;
        for (int t = 0; t < 4; t++) o[t] = oneant_obs_lane(C, L[t], S[t], t, origin, act0[t], act1[t], &sens[6 * t], s_obs.data(), core, pg);
        float e[4] = {o[0].ec, o[1].ec, o[2].ec, o[3].ec}, l[4] = {o[0].lim, o[1].lim, o[2].lim, o[3].lim}, c[4] = {o[0].acost, o[1].acost, o[2].acost, o[3].acost};
        if (do_physics && reset_flag == 0)
            for (int t = 0; t < 4; t++) for (int i = 0; i < 6; i++) b.foot_sensors[(size_t)env * 24 + 6 * t + i] = sens[6 * t + i];
        float pbx = prev_env[0], pby = prev_env[1], bbx = prev_env[2], bby = prev_env[3];
        float tbx = 0.f - bgx, tby = 0.f - bgy;
        float pot = -sqrtf(tbx * tbx + tby * tby + 0.f * 0.f) / C->dt;
        oneant_reward(C, pg.z, core.up_proj, quad4(e), quad4(l), quad4(c), pbx, pby, bbx, bby, pg.x, pg.y, bgx, bgy, B.qx, B.qy, B.qz, B.qw, progress, rew, rs);
        prev_env[0] = pg.x; prev_env[1] = pg.y; prev_env[2] = bgx; prev_env[3] = bgy; prev_env[4] = pot; prev_env[5] = pot_in;
    }
    b.rew[env] = rew; b.reset[env] = rs; b.progress[env] = progress;
    if (b.rew_out) b.rew_out[env] = rew;
    if (b.done_out) b.done_out[env] = (uint8_t)rs;
    for (int i = 0; i < obs_dim; i++) {
        const float c = clampf(s_obs[i], -C->clip_obs, C->clip_obs);
        if (b.obs) b.obs[(size_t)env * obs_dim + i] = s_obs[i];
        if (b.obs_clipped) b.obs_clipped[(size_t)env * obs_dim + i] = c;
        if (b.obs_out) b.obs_out[(size_t)env * obs_dim + i] = c;
    }
    if (b.obs_planes) {                                                 // as the epilogue of ant_step_kernel
        const int KC = (obs_dim + 31) / 32;
        for (int k = 0; k < KC * 32; k++) {
            const float t = k < obs_dim ? clampf(s_obs[k], -C->clip_obs, C->clip_obs) * b.obs_planes_scale : 0.f;
            uint16_t* c = b.obs_planes + ((size_t)env * KC + k / 32) * 64 + k % 32;
            c[0] = f2h(t);
            c[32] = f2h((t - h2f(c[0])) * 2048.f);
        }
    }
}

static void host_heli_env(const mms_config* C, const HostBufs& b, int env, int do_physics) {
    const mms_model* M = &C->model;
    const int A = 4;
    int64_t reset_flag = b.reset[env], progress = b.progress[env];
    float rows[4][13];
    for (int k = 0; k < A; k++) {
        float* root = b.root_states + ((size_t)env * A + k) * 13;
        const float* init = b.initial_root_states + ((size_t)env * A + k) * 13;
        float* dof = b.dof_state + ((size_t)env * A + k) * 8;
        const float* act = b.actions + (size_t)env * 24 + 6 * k;
        RigidState B = load_rigid(root);
        if (do_physics && reset_flag == 0) {
            V3 thr[2];
            for (int r = 0; r < 2; r++) {
                float a0 = clampf(act[3 * r], -C->clip_actions, C->clip_actions), a1 = clampf(act[3 * r + 1], -C->clip_actions, C->clip_actions),
                      a2 = clampf(act[3 * r + 2], -C->clip_actions, C->clip_actions);
                float tz = C->dt * clampf(a2 * 2000.f, -2000.f, 2000.f);
                thr[r] = V3{tz * clampf(a0, -0.2f, 0.2f), tz * clampf(a1, -0.2f, 0.2f), tz};
            }
            float h = C->dt / (float)C->substeps;
            for (int s = 0; s < C->substeps; s++) {
                heli_substep(M, h, B, thr[0], thr[1]);
                for (int j = 0; j < 4; j++) dof[2 * j] += h * dof[2 * j + 1];
            }
        }
        if (reset_flag != 0) {
            B = load_rigid(init);
            for (int j = 0; j < 4; j++) { dof[2 * j] = 0.f; dof[2 * j + 1] = (j == 1) ? -50.f : (j == 3 ? 50.f : 0.f); }
        }
        store_rigid(root, B);
        store_rigid(rows[k], B);
        rows[k][0] += b.env_origin[3 * env]; rows[k][1] += b.env_origin[3 * env + 1]; rows[k][2] += b.env_origin[3 * env + 2];
        for (int j = 0; j < 13; j++) {
            const float c = clampf(rows[k][j], -C->clip_obs, C->clip_obs);
            if (b.obs) b.obs[(size_t)env * 52 + 13 * k + j] = rows[k][j];
            if (b.obs_clipped) b.obs_clipped[(size_t)env * 52 + 13 * k + j] = c;
            if (b.obs_out) b.obs_out[(size_t)env * 52 + 13 * k + j] = c;
        }
    }
    progress += 1;
    if (reset_flag != 0) { progress = 0; b.reset_count[env] += 1; }
    float rew;
    int64_t rs;
    ingenuity_reward(&rows[0][0], C->max_episode_length, progress, rew, rs);
    b.rew[env] = rew; b.reset[env] = rs; b.progress[env] = progress;
    if (b.rew_out) b.rew_out[env] = rew;
    if (b.done_out) b.done_out[env] = (uint8_t)rs;
}


}  // namespace mms
