// maze.hip — batched MetaMaze engine for gfx950.
//
// Replaces, for N environments per launch, the reference's per-object Python/numba hot path
// (paths relative to metagym/metamaze/envs):
//   MazeBase.reset / evaluation_rule          maze_base.py:40-63, :65-95        (reset_env, eval_*)
//   MazeCore2D.do_action / update_observation maze_2d.py:21-34, :89-121         (maze2d_step_kernel)
//   MazeCoreDiscrete3D.turn / move            maze_discrete_3d.py:51-72         (transition_discrete)
//   vector_move_with_collision & friends      dynamics.py:17-92                 (transition_continuous)
//   DDA_2D / maze_view                        ray_caster_utils.py:11-62, :66-209 (column_pass, pixel_pass)
//   life bar overlay                          maze_discrete_3d.py:118-126       (pixel_pass)
//
// Design (MI355X)
//   * 2-D maze: integer grid logic + a (2v+1)^2 window -> one lane per env, SoA state.
//   * 3-D mazes: the cost is the first-person image (12*H*V bytes of int32 per env-step, 786 KB at
//     256x256), so one 256-thread workgroup renders one env. The task's wall / texture-id /
//     translucency grids are staged in LDS. Rendering is two passes per wave over its 64 screen
//     columns: (A) lane = column: column direction table + DDA grid traversal, results (wall span,
//     texture row, fog, the ordered list of translucent cells crossed) parked in wave-private LDS;
//     (B) lane = screen row: for each of the 64 columns every lane shades one pixel — floor or
//     ceiling cast, wall overwrite, translucent blends in ray order — and the wave stores 64
//     consecutive int32x3 pixels (768 contiguous bytes) per instruction. The reference paints
//     floor, ceiling, walls and overlays in four sequential sweeps over an int32 frame buffer;
//     because every sweep only touches the pixel it is visiting, the per-pixel composition here
//     performs the same operations in the same order and is bit-identical.
//   * Arithmetic mirrors numba's typing of the reference (float64 everywhere except the float32
//     column tables and texture values; every frame-buffer store truncates toward zero); the file
//     is compiled with -ffp-contract=off. Divisions by cell_size / text_size / text-to-cell ratio
//     become multiplications only when the divisor is a power of two (bit-identical).
//   * No MFMA (no contraction), no atomics, no inter-workgroup communication.
#include <cstdlib>
#include <cstring>

#include <atomic>
#include <mutex>
#include <vector>

#include "mg_common.h"

namespace {

constexpr int MZ_BLOCK = 256;
constexpr int MZ_WAVES = MZ_BLOCK / mg::WAVE;
constexpr int SLAB = 32;   // default screen columns a wave handles per pass (ViewK::slab, <= 64); sets the per-wave overlay-record LDS

// ------------------------------------------------------------------------------------------------
// task / state access
// ------------------------------------------------------------------------------------------------

struct Task {   // one row of the task table, resolved for an env
    int n, nn;
    const int8_t *walls;
    const uint8_t *texts;
    const double *food;
    const int32_t *interval;
    const int16_t *food_cells;   // optional compact list of the cells that can hold food (NULL = all cells)
    int n_food;
    const int16_t *cell_slot;    // (slot layout) inverse of food_cells for this task: slot of a cell, -1 = never holds food
    const double *slot_food;     // (slot layout) [k * slot_stride]: food_rewards of the task's k-th listed cell
    const int32_t *slot_interval;
    int slot_stride;
    int sx, sy, gx, gy;
    double cell_size, wall_h, agent_h, init_life, max_life, step_reward, goal_reward;
};

__device__ __forceinline__ Task load_task(const mg_maze_tasks &T, int tid) {
    Task t;
    t.n = T.n;
    t.nn = T.n * T.n;
    const size_t off = (size_t)tid * t.nn;
    t.walls = T.walls + off;
    t.texts = T.texts + off;
    t.food = T.food_rewards ? T.food_rewards + off : nullptr;
    t.interval = T.food_interval ? T.food_interval + off : nullptr;
    t.food_cells = (T.food_cells != nullptr && T.n_food != nullptr) ? T.food_cells + (size_t)tid * T.max_food : nullptr;
    t.n_food = t.food_cells ? T.n_food[tid] : t.nn;
    t.cell_slot = T.cell_slot ? T.cell_slot + off : nullptr;
    t.slot_food = T.slot_food ? T.slot_food + tid : nullptr;
    t.slot_interval = T.slot_interval ? T.slot_interval + tid : nullptr;
    t.slot_stride = T.n_tasks;
    t.sx = T.start[2 * tid];
    t.sy = T.start[2 * tid + 1];
    t.gx = T.goal[2 * tid];
    t.gy = T.goal[2 * tid + 1];
    const double *s = T.scalars + (size_t)tid * 8;
    t.cell_size = s[0]; t.wall_h = s[1]; t.agent_h = s[2]; t.init_life = s[3];
    t.max_life = s[4]; t.step_reward = s[5]; t.goal_reward = s[6];
    return t;
}

struct Agent {   // scalar per-env state
    int gx, gy, steps, ori_idx;
    double ori, life;
    float lx, ly;
};

__device__ __forceinline__ Agent load_agent(const mg_maze_state &st, int n_envs, int e) {
    Agent a;
    a.gx = st.grid[e];
    a.gy = st.grid[n_envs + e];
    a.steps = st.steps[e];
    a.ori_idx = st.ori_idx ? st.ori_idx[e] : 0;
    a.ori = st.ori ? st.ori[e] : 0.0;
    a.lx = st.loc ? st.loc[e] : 0.0f;
    a.ly = st.loc ? st.loc[n_envs + e] : 0.0f;
    a.life = st.life ? st.life[e] : 0.0;
    return a;
}

__device__ __forceinline__ void store_agent(const mg_maze_state &st, int n_envs, int e, const Agent &a) {
    st.grid[e] = a.gx;
    st.grid[n_envs + e] = a.gy;
    st.steps[e] = a.steps;
    if (st.ori_idx) st.ori_idx[e] = a.ori_idx;
    if (st.ori) st.ori[e] = a.ori;
    if (st.loc) { st.loc[e] = a.lx; st.loc[n_envs + e] = a.ly; }
    if (st.life) st.life[e] = a.life;
}

// MazeBase.reset maze_base.py:40-63 — scalar part
__device__ __forceinline__ void reset_agent(const Task &t, int task_type, Agent &a) {
    a.gx = t.sx;
    a.gy = t.sy;
    a.lx = (float)(t.sx * t.cell_size + 0.5 * t.cell_size);   // get_cell_center maze_base.py:194-197
    a.ly = (float)(t.sy * t.cell_size + 0.5 * t.cell_size);
    a.ori = 0.0;
    a.ori_idx = 0;
    a.steps = 0;
    if (task_type == MG_MAZE_SURVIVAL) a.life = t.init_life;
}

// index of cell c of env e in the SURVIVAL arrays: [N][n*n] (cell_stride 1) for the workgroup-per-env
// 3-D kernel, [n*n][N] (env_stride 1) for the lane-per-env 2-D kernel, so both access them coalesced
__device__ __forceinline__ size_t fidx(const mg_maze_state &st, int e, int c) {
    return (size_t)e * st.food_env_stride + (size_t)c * st.food_cell_stride;
}

// MazeBase.reset — per-cell part (SURVIVAL), cells c = first, first+stride, ...
// full = true (mg_maze_reset): every cell is written. full = false (fused auto-reset): with a food-cell list
// only those cells are rewritten — the others still hold the values the last full reset gave them, because
// nothing ever changes a cell whose food is <= 1e-2 and whose interval is 0.
__device__ __forceinline__ void reset_cells(const Task &t, const mg_maze_state &st, int e, int first, int stride,
                                            bool full) {
    if (st.food_by_slot) {            // slot layout: the listed cells are all there is
        for (int k = first; k < t.n_food; k += stride) {
            const size_t i = fidx(st, e, k);
            st.wait_refresh[i] = 0;
            st.cur_food[i] = t.slot_food[(size_t)k * t.slot_stride];
            st.revival[i] = t.slot_interval[(size_t)k * t.slot_stride];
        }
        return;
    }
    const bool listed = !full && t.food_cells != nullptr;
    const int count = listed ? t.n_food : t.nn;
    for (int k = first; k < count; k += stride) {
        const int c = listed ? (int)t.food_cells[k] : k;
        const size_t i = fidx(st, e, c);
        st.wait_refresh[i] = 0;
        st.cur_food[i] = t.food[c];
        st.revival[i] = t.interval[c];
    }
}

// evaluation_rule maze_base.py:65-95 — scalar part. Returns done; touches the agent's cell only.
__device__ __forceinline__ int eval_scalar(const Task &t, const mg_maze_state &st, int e, int task_type,
                                           int max_steps, Agent &a, double &reward) {
    a.steps += 1;
    if (task_type == MG_MAZE_SURVIVAL) {
        int c = a.gx * t.n + a.gy;
        if (st.food_by_slot) c = t.cell_slot[c];            // a cell outside the list never holds more than 1e-2: nothing to eat
        const size_t g = c >= 0 ? fidx(st, e, c) : 0;
        double r = 0.0;
        const double f = c >= 0 ? st.cur_food[g] : 0.0;
        if (f > 1.0e-2) {                                   // :71-75
            r = f;
            st.wait_refresh[g] = 1;
            st.cur_food[g] = 0.0;
        }
        a.life += r + t.step_reward;                        // :78
        if (t.max_life < a.life) a.life = t.max_life;       // :79
        reward = r;
        return (a.life < 0.0) || (a.steps > max_steps - 1); // :80, :191-192
    }
    const int goal = (t.gx == a.gx) && (t.gy == a.gy);
    reward = t.step_reward + goal * t.goal_reward;          // :92
    return goal || (a.steps > max_steps - 1);
}

// evaluation_rule :83-88 — food revival over the cells c = first, first+stride, ...
// A cell whose task entry has food_interval == 0 can never hold food (food_interval =
// interval * (food_rewards > 1e-3), maze_task.py:172): its wait flag stays 0 and its counter stays
// 0, so skipping it is exact and saves the HBM round trip for the ~95 % of cells that are empty.
__device__ __forceinline__ void eval_cells(const Task &t, const mg_maze_state &st, int e, int first, int stride) {
    if (st.food_by_slot) {
        // Only a WAITING slot changes: revival -= wait leaves a slot that is not waiting as it is, and its counter cannot be
        // negative (it is >= 0 after every reset and renewal and only counts down while waiting; mg_maze_state documents the
        // invariant and load_state_dict checks it). So the common case is one byte read per slot — no counter read, no store.
        for (int k = first; k < t.n_food; k += stride) {
            const size_t i = fidx(st, e, k);
            if (st.wait_refresh[i] == 0) continue;
            int rv = st.revival[i] - 1;
            if (rv < 0) {
                st.cur_food[i] = t.slot_food[(size_t)k * t.slot_stride];
                rv = t.slot_interval[(size_t)k * t.slot_stride];
                st.wait_refresh[i] = 0;
            }
            st.revival[i] = rv;
        }
        return;
    }
    for (int k = first; k < t.n_food; k += stride) {
        const int c = t.food_cells ? (int)t.food_cells[k] : k;
        const int interval = t.interval[c];
        if (interval == 0 && !(t.food[c] > 1.0e-2)) continue;
        const size_t i = fidx(st, e, c);
        int rv = st.revival[i] - (int)st.wait_refresh[i];
        if (rv < 0) {
            st.cur_food[i] = t.food[c];
            rv = interval;
            st.wait_refresh[i] = 0;
        }
        st.revival[i] = rv;
    }
}

// ------------------------------------------------------------------------------------------------
// MetaMaze2D: one lane per env
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(MZ_BLOCK) void maze2d_step_kernel(mg_maze_tasks T, mg_maze_state st, int task_type,
                                                               int max_steps, int vg, int auto_reset, int n_envs,
                                                               const int32_t *action, float *obs, float *reward,
                                                               double *reward64, uint8_t *done) {
    const int e = blockIdx.x * MZ_BLOCK + threadIdx.x;
    if (e >= n_envs) return;
    const Task t = load_task(T, st.task_id[e]);
    Agent a = load_agent(st, n_envs, e);
    if (action != nullptr) {
        const int act = action[e] & 3;
        // DISCRETE_ACTIONS maze_env.py:14 = [(-1,0),(1,0),(0,-1),(0,1)]; maze_2d.py:24-29
        const int ti = a.gx + (act == 0 ? -1 : (act == 1 ? 1 : 0));
        const int tj = a.gy + (act == 2 ? -1 : (act == 3 ? 1 : 0));
        const int wi = ti < 0 ? ti + t.n : ti, wj = tj < 0 ? tj + t.n : tj;   // python negative index
        if (wi < t.n && wj < t.n && t.walls[wi * t.n + wj] < 1) { a.gx = ti; a.gy = tj; }
        double r;
        int d = eval_scalar(t, st, e, task_type, max_steps, a, r);
        if (task_type == MG_MAZE_SURVIVAL) eval_cells(t, st, e, 0, 1);
        if (reward) reward[e] = (float)r;
        if (reward64) reward64[e] = r;
        done[e] = (uint8_t)d;
        if (d && auto_reset) {
            reset_agent(t, task_type, a);
            if (task_type == MG_MAZE_SURVIVAL) reset_cells(t, st, e, 0, 1, false);
        }
        store_agent(st, n_envs, e, a);
    }
    // update_observation maze_2d.py:89-121
    const int w = 2 * vg + 1;
    float *o = obs + (size_t)e * w * w;
    for (int p = 0; p < w; ++p)
        for (int q = 0; q < w; ++q) {
            const int x = a.gx - vg + p, y = a.gy - vg + q;
            float v = -1.0f;
            if (x >= 0 && x < t.n && y >= 0 && y < t.n) {
                v = (float)(-(int)t.walls[x * t.n + y]);                                   // :113
                if (task_type == MG_MAZE_SURVIVAL) {                                       // :117
                    double food;
                    if (st.food_by_slot) {
                        // (-1: not a food cell and its task value is exactly 0.0 — nothing to read; -2: not a food cell, a
                        // nonzero value <= 1e-2 that never changes)
                        const int k = t.cell_slot[x * t.n + y];
                        food = k >= 0 ? st.cur_food[fidx(st, e, k)] : (k == -1 ? 0.0 : t.food[x * t.n + y]);
                    } else food = st.cur_food[fidx(st, e, x * t.n + y)];
                    v = (float)((double)v + food);
                }
                else v = (float)((double)v + ((x == t.gx && y == t.gy) ? 1.0 : 0.0));       // :120
            }
            o[p * w + q] = v;
        }
    if (task_type == MG_MAZE_SURVIVAL) o[vg * w + vg] = (float)a.life;                      // :118
}

// ------------------------------------------------------------------------------------------------
// 3-D transitions (executed by one thread of the env's workgroup)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ void transition_discrete(const Task &t, int act, Agent &a) {
    const int turn = act == 0 ? -1 : (act == 1 ? 1 : 0);      // maze_env.py:14
    const int step = act == 2 ? -1 : (act == 3 ? 1 : 0);
    a.ori_idx = (a.ori_idx + turn) & 3;                        // maze_discrete_3d.py:69-72
    int g0 = a.gx, g1 = a.gy;                                  // :51-67
    if (a.ori_idx == 0) g0 += step;
    else if (a.ori_idx == 1) g1 += step;
    else if (a.ori_idx == 2) g0 -= step;
    else g1 -= step;
    if (g0 >= 0 && g0 < t.n && g1 >= 0 && g1 < t.n && t.walls[g0 * t.n + g1] == 0) { a.gx = g0; a.gy = g1; }
}

__device__ __forceinline__ float nearest_point(const float *pos, const float *l1, const float *l2, float *np_out) {
    // dynamics.py:17-29
    float u0 = l2[0] - l1[0], u1 = l2[1] - l1[1];
    const float edge_norm = sqrtf(u0 * u0 + u1 * u1);
    const double m = (1.0e-6 > (double)edge_norm) ? 1.0e-6 : (double)edge_norm;
    u0 = (float)((double)u0 / m);
    u1 = (float)((double)u1 / m);
    const float dist_1 = (pos[0] - l1[0]) * u0 + (pos[1] - l1[1]) * u1;
    float p0, p1;
    if (dist_1 > edge_norm) { p0 = l2[0]; p1 = l2[1]; }
    else if (dist_1 < 0) { p0 = l1[0]; p1 = l1[1]; }
    else { p0 = l1[0] + dist_1 * u0; p1 = l1[1] + dist_1 * u1; }
    const float d0 = pos[0] - p0, d1 = pos[1] - p1;
    np_out[0] = p0;
    np_out[1] = p1;
    return sqrtf(d0 * d0 + d1 * d1);
}

__device__ __forceinline__ void collision_force(const float *dv, double cell_size, double col_dist, float *out) {
    // dynamics.py:32-56
    const double dist = (double)sqrtf(dv[0] * dv[0] + dv[1] * dv[1]);
    const double eff = col_dist / cell_size;
    out[0] = out[1] = 0.0f;
    if (dist > 0.708 + eff) return;
    if (fabsf(dv[0]) < 0.5f && fabsf(dv[1]) < 0.5f) {
        const double mx = dist > 1.0e-6 ? dist : 1.0e-6;
        const float k = (float)(0.50 / mx * (0.708 + eff - dist) * cell_size);
        out[0] = k * dv[0];
        out[1] = k * dv[1];
        return;
    }
    const bool x_pos = (dv[0] + dv[1]) > 0, y_pos = (dv[1] - dv[0]) > 0;
    const float A[2] = {0.5f, 0.5f}, B[2] = {-0.5f, 0.5f}, C[2] = {-0.5f, -0.5f}, D[2] = {0.5f, -0.5f};
    float np_[2], d;
    if (x_pos && y_pos) d = nearest_point(dv, A, B, np_);
    else if (!x_pos && y_pos) d = nearest_point(dv, B, C, np_);
    else if (!x_pos && !y_pos) d = nearest_point(dv, C, D, np_);
    else d = nearest_point(dv, D, A, np_);
    if (eff < (double)d) return;
    float o0 = dv[0] - np_[0], o1 = dv[1] - np_[1];
    const float on = sqrtf(o0 * o0 + o1 * o1);
    const double r = 1.0 / ((1.0e-6 > (double)on) ? 1.0e-6 : (double)on);
    o0 = (float)((double)o0 * r);
    o1 = (float)((double)o1 * r);
    const float k = (float)(0.50 * (eff - (double)d) * cell_size);
    out[0] = k * o0;
    out[1] = k * o1;
}

__device__ __forceinline__ void transition_continuous(const Task &t, double col_dist, float turn, float walk,
                                                      Agent &a) {
    // maze_continuous_3d.py:47-56 + dynamics.py:59-92
    const double PI = 3.1415926, t_PI = 6.2831852;
    const double tr = (double)turn, ws = (double)walk;
    const double turn_rate = (tr < -1 ? -1 : (tr > 1 ? 1 : tr)) * PI;
    double walk_speed = ws < -1 ? -1 : (ws > 1 ? 1 : ws);
    if (walk_speed < 0) walk_speed *= 0.50;
    double ori = a.ori;
    float p0 = a.lx, p1 = a.ly;
    const float cs32 = (float)t.cell_size;
    for (int it = 0; it < 10; ++it) {   // int(100 * 0.10)
        double fin = ori + turn_rate * 0.01;
        const double off_ori = 0.5 * (fin + ori);
        const double off = walk_speed * 0.01;
        const float d_x = (float)(cos(off_ori) * off), d_y = (float)(sin(off_ori) * off);
        while (fin > t_PI) fin -= t_PI;
        while (fin < 0) fin += t_PI;
        ori = fin;
        const float e0 = p0 + d_x, e1 = p1 + d_y;
        const float c0 = e0 / cs32, c1 = e1 / cs32;
        float col0 = 0.0f, col1 = 0.0f;
        for (int i = -1; i < 2; ++i)
            for (int j = -1; j < 2; ++j) {
                const int w_i = i + (int)c0, w_j = j + (int)c1;
                if (w_i > -1 && w_i < t.n && w_j > -1 && w_j < t.n && t.walls[w_i * t.n + w_j] > 0) {
                    float cd[2], f[2];
                    cd[0] = (c0 - floorf(c0)) - ((float)i + 0.5f);
                    cd[1] = (c1 - floorf(c1)) - ((float)j + 0.5f);
                    collision_force(cd, t.cell_size, col_dist, f);
                    col0 += f[0];
                    col1 += f[1];
                }
            }
        p0 = col0 + e0;
        p1 = col1 + e1;
    }
    a.ori = ori;
    a.lx = p0;
    a.ly = p1;
    a.gx = (int)(p0 / cs32);   // get_loc_grid maze_base.py:199-202
    a.gy = (int)(p1 / cs32);
}

// ------------------------------------------------------------------------------------------------
// renderer
// ------------------------------------------------------------------------------------------------

struct ViewK {
    int H, V, TS, t_max, obs_u8;
    int slab;          // screen columns a wave handles per pass (32 or 64)
    double max_vision, max_vision_lo, inv_max_vision, l_focal, text_size, inv_text_size, half_h, half_v, pixel_size;
    double col_dist;
    int text_size_pow2;
    double eff_max;    // bound on the floor / ceiling cast distance: max_vision / cos(fov / 2)
    const double *col_cos, *col_sin;
    float ori_sin[4], ori_cos[4];
    const uint32_t *tex, *ceil_tex;
    long long ceil_delta;   // ceil_tex - tex in texels: the cast selects an OFFSET from the one base `tex`, not one of two pointers
};

// int(x) truncation with python's unbounded ints replaced by clamping outside [lo-1, hi+1]
__device__ __forceinline__ int to_int_clamped(double x, int lo, int hi) {
    if (!(x > (double)lo - 1.0)) return lo;
    if (x >= (double)hi + 1.0) return hi + 1;
    return (int)x;
}
// int(x) for a cell index: v_cvt_i32_f64 truncates toward zero and SATURATES out-of-range inputs (NaN -> 0),
// so one instruction gives python's int() wherever the result can matter (|x| < 2^31) and a harmless
// out-of-grid value elsewhere; `(unsigned)i < n` then tests 0 <= i < n with one compare.
__device__ __forceinline__ int cell_index(double x) { return __double2int_rz(x); }

struct EnvShared {   // one per workgroup, LDS
    Agent a;
    double reward;
    int done;
    double pos[2], s_ori, c_ori;
};

// Column record produced by pass A. Each lane keeps the record of ITS column in registers; pass B
// broadcasts column k's record to the whole wave with v_readlane (SGPR operands) — no LDS round trip.
// a / b, correctly rounded, from rb = RN(1/b) computed once (Markstein 1990: q = RN(a*rb), exact residual
// r = a - b*q by FMA, result RN(q + r*rb) == RN(a/b) for normal-range operands whose divisor's significand
// is not all ones — here b is a float32-valued cos_hp or the constant max_vision). Bit-identical to the
// IEEE division the reference performs, at 3 VALU ops instead of ~28 per pixel.
__device__ __forceinline__ double div_by(double a, double b, double rb) {
    const double q = a * rb;
    const double r = fma(-b, q, a);
    return fma(r, rb, q);
}

// texel channels of a packed r | g<<8 | b<<16 word as doubles: v_cvt_f32_ubyteN + v_cvt_f64_f32 (exact)
__device__ __forceinline__ double tex_r(uint32_t tx) {
    float f; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(tx)); return (double)f;
}
__device__ __forceinline__ double tex_g(uint32_t tx) {
    float f; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(tx)); return (double)f;
}
__device__ __forceinline__ double tex_b(uint32_t tx) {
    float f; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(tx)); return (double)f;
}

struct ColRec {
    float cos_hp, cos_abs, sin_abs;
    double rcos_hp;    // RN(1 / cos_hp)
    double w_oma;      // 1 - alpha of the wall hit
    double w_ratio;    // hit_dist * cos_hp / l_focal
    float w_light;     // |cos| or |sin| (a float32 value in the reference too)
    int w_tex;         // texel offset of (texture id, texture row)
    int w_span;        // v_s | v_e << 12 | n_tr << 24   (v_s >= v_e: no wall / beyond max_vision;
                       // n_tr = number of translucent records of the column)
};

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double bcast(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ ColRec bcast(const ColRec &r, int lane) {
    ColRec o;
    o.cos_hp = bcast(r.cos_hp, lane); o.cos_abs = bcast(r.cos_abs, lane); o.sin_abs = bcast(r.sin_abs, lane);
    o.rcos_hp = bcast(r.rcos_hp, lane);
    o.w_oma = bcast(r.w_oma, lane); o.w_ratio = bcast(r.w_ratio, lane); o.w_light = bcast(r.w_light, lane);
    o.w_tex = bcast(r.w_tex, lane); o.w_span = bcast(r.w_span, lane);
    return o;
}

// The record as the pixel pass consumes it: the float32 table values widened once per column, not once per pixel.
struct ColRecD {
    double cos_hp, cos_abs, sin_abs, rcos_hp, w_oma, w_ratio, w_light;
    int w_tex, w_span;
};
__device__ __forceinline__ ColRecD widen(const ColRec &r) {
    return ColRecD{(double)r.cos_hp, (double)r.cos_abs, (double)r.sin_abs, r.rcos_hp, r.w_oma, r.w_ratio, (double)r.w_light, r.w_tex, r.w_span};
}
// SMALL kernels: pass A leaves each column's widened record in LDS (64 bytes, one slot per lane of the slab) and the pixel pass
// reads slot k with four ds_read_b128 at ONE address for the wave — LDS broadcasts, no bank conflict, no VALU issue slot (the
// general kernel's 13 v_readlane + 4 converts per column are VALU work, and at one 64-row chunk per column the 64 x 64 kernel is
// VALU-bound: 82 % busy, profiles/r05/pmc_maze3d_64x64.json). The values arrive in VGPRs, uniform across the wave; only the span
// word, which steers scalar branches, is also lifted into an SGPR. (ds_bpermute_b32 broadcasts were tried first: 15 of them per
// column saturate the LDS pipe — 106 % busy — and the kernel got 6 % slower.)
struct __attribute__((aligned(16))) ColRecLds { double cos_hp, cos_abs, sin_abs, rcos_hp, w_oma, w_ratio, w_light; int w_tex, w_span; };
static_assert(sizeof(ColRecLds) == 64, "four ds_read_b128");

// Pass A, lane = screen column: ray_caster_utils.py:84-90 (direction tables), :11-62 (DDA_2D) and the
// per-column parts of :155-205.
template <int REC, bool STOCK>
__device__ ColRec column_pass(const ViewK &vk, const Task &t, const EnvShared &es, const int8_t *walls,
                              const uint8_t *texts, const double *transp, int col, int lane,
                              uint32_t *entries /* [t_max][SLAB] records of REC words */, double cs, double inv_cs,
                              int cs_pow2) {
    const int n = t.n;
    const double chp = vk.col_cos[col], shp = vk.col_sin[col];
    const float sin_abs = (float)(shp * es.c_ori + chp * es.s_ori);
    const float cos_abs = (float)(chp * es.c_ori - shp * es.s_ori);
    const float cos_hp = (float)chp;

    const double vh = t.agent_h, ceil_h = t.wall_h;
    const double px0 = es.pos[0], px1 = es.pos[1];
    const int i0 = (int)(cs_pow2 ? px0 * inv_cs : px0 / cs), j0 = (int)(cs_pow2 ? px1 * inv_cs : px1 / cs);
    const double c = (double)cos_abs, s = (double)sin_abs;
    const bool cz = fabs(c) < 1.0e-6, sz = fabs(s) < 1.0e-6;
    const double delta_x = cz ? 1.0e+6 : fabs(cs / c);
    const double delta_y = sz ? 1.0e+6 : fabs(cs / s);
    const double d_x = c > 0 ? ((i0 + 1) * cs - px0) : (i0 * cs - px0);
    const double d_y = s > 0 ? ((j0 + 1) * cs - px1) : (j0 * cs - px1);
    double side_x = cz ? 1.0e+6 : d_x / c;
    double side_y = sz ? 1.0e+6 : d_y / s;
    const int di = c > 0 ? 1 : -1, dj = s > 0 ? 1 : -1;
    int hi = i0, hj = j0, side = 0, n_tr = 0;
    double hit_dist = 0.0;

    // translucent record: screen span of the cell boundary at distance `dist` (:195-203)
    auto add_record = [&](double dist, int cell) {
        if (n_tr >= vk.t_max) return;
        const double r2 = dist * (double)cos_hp / vk.l_focal;
        const double tv = (ceil_h - vh) / r2, bv = vh / r2;
        int s2 = to_int_clamped((vk.half_v - tv) / vk.pixel_size, 0, vk.V);
        int e2 = to_int_clamped((vk.half_v + bv) / vk.pixel_size, -1, vk.V - 1);
        if (s2 < 0) s2 = 0;
        if (e2 > vk.V) e2 = vk.V;
        // range(v_s, v_e) of the reference is empty here (also for the -1 that stands for a negative / NaN bound): no pixel
        // can be touched, and a negative e2 must not reach the packed words below (its sign bits would overwrite the cell field)
        if (e2 <= s2) return;
        // compact record (one word: 12-bit span bounds, 8-bit cell) when the frame has < 4096 rows and the maze
        // <= 256 cells — half the LDS of the overlay records, which is what bounds the resident envs per CU
        if (REC == 1) entries[n_tr * vk.slab + lane] = (unsigned)s2 | ((unsigned)e2 << 12) | ((unsigned)cell << 24);
        else {
            entries[(n_tr * vk.slab + lane) * 2] = (unsigned)s2 | ((unsigned)e2 << 16);
            entries[(n_tr * vk.slab + lane) * 2 + 1] = (unsigned)cell;
        }
        ++n_tr;
    };

    if (hi >= 0 && hi < n && hj >= 0 && hj < n && transp[hi * n + hj] > 0.01)   // :25-29
        add_record(side_x < side_y ? side_x : side_y, hi * n + hj);
    while (hit_dist < vk.max_vision) {                                           // :31-61
        const bool xs = side_x < side_y;
        if (xs) { hi += di; side_y -= side_x; hit_dist += side_x; }
        else { hj += dj; side_x -= side_y; hit_dist += side_y; }
        if (hi < 0 || hi >= n) {
            if (hj < 0 || hj >= n) { hit_dist = 1.0e+6; break; }
        } else if (hj >= 0 && hj < n) {
            const int cell = hi * n + hj;
            if (transp[cell] > 0.01) add_record(hit_dist, cell);
            if (walls[cell] > 0) { side = xs ? 0 : 1; break; }
        }
        if (xs) side_x = delta_x;
        else side_y = delta_y;
    }

    int span = 0;   // empty
    double oma = 0.0, ratio = 1.0;
    float light = 0.0f;
    int texoff = 0;
    const int ci = hi < 0 ? hi + n : hi, cj = hj < 0 ? hj + n : hj;
    if (!(hit_dist > vk.max_vision) && ci >= 0 && ci < n && cj >= 0 && cj < n) {   // :160-161
        double alpha = 2.0 * hit_dist / vk.max_vision - 1.0;
        alpha = alpha > 0.0 ? alpha : 0.0;
        alpha = alpha < 1.0 ? alpha : 1.0;
        oma = 1.0 - alpha;
        const int text_id = texts[ci * n + cj];
        const double hpx = hit_dist * c + px0, hpy = hit_dist * s + px1;
        double local_h = side == 0 ? hpy : hpx;                                    // :166-173
        local_h = cs_pow2 ? local_h * inv_cs : local_h / cs;
        local_h -= floor(local_h);
        light = fabsf(side == 0 ? cos_abs : sin_abs);
        ratio = hit_dist * (double)cos_hp / vk.l_focal;                            // :175
        const double top_v = (ceil_h - vh) / ratio, bot_v = vh / ratio;
        int v_s = to_int_clamped((vk.half_v - top_v) / vk.pixel_size, 0, vk.V);
        int v_e = to_int_clamped((vk.half_v + bot_v) / vk.pixel_size, -1, vk.V - 1);
        if (v_s < 0) v_s = 0;
        if (v_e > vk.V) v_e = vk.V;
        span = v_s | (v_e << 12);
        double d_i = (STOCK || vk.text_size_pow2) ? local_h * vk.inv_text_size : local_h / vk.text_size;   // :184-188
        d_i -= floor(d_i);
        const int ti = (int)(vk.TS * d_i);
        texoff = (text_id * vk.TS + ti) * vk.TS;
    } else {
        n_tr = 0;   // `continue` at :160-161 also skips the overlays of this column
    }
    ColRec rec;
    rec.cos_hp = cos_hp; rec.cos_abs = cos_abs; rec.sin_abs = sin_abs;
    rec.rcos_hp = 1.0 / (double)cos_hp;
    rec.w_oma = oma; rec.w_ratio = ratio; rec.w_light = light;
    rec.w_tex = texoff; rec.w_span = span | (n_tr << 24);
    return rec;
}

struct RowK {   // per-lane constants of a screen row (they do not depend on the column)
    int kind;   // 0 untouched, 1 floor, 2 ceiling
    double distance, light, ys;
};
// the same as one 32-byte LDS row (two ds_read_b128 from ONE address; the table is padded to whole 64-row chunks with
// kind-0 rows, so the pixel loop needs neither a clamp for the ragged last chunk nor a second table for `kind`)
struct __attribute__((aligned(16))) RowRec { double distance, light, ys; int kind, pad; };

__device__ __forceinline__ RowK row_constants(const ViewK &vk, const Task &t, int d_v) {
    RowK r;
    r.kind = 0;
    r.distance = 0.0;
    r.light = 0.0;
    const double yc = (d_v + 0.5) * vk.pixel_size;
    r.ys = vk.half_v - yc;                                     // used by the wall texture lookup (:182)
    if (d_v > vk.V / 2) {                                      // floor rows :95-101
        const double v_screen = yc - vk.half_v;
        r.distance = t.agent_h / v_screen * vk.l_focal;
        r.light = v_screen / vk.l_focal;
        r.kind = r.distance > vk.max_vision ? 0 : 1;
    } else if (d_v < vk.V / 2) {                               // ceiling rows :129-134
        const double v_screen = vk.half_v - yc;
        r.distance = (t.wall_h - t.agent_h) / v_screen * vk.l_focal;
        r.light = v_screen / vk.l_focal;
        r.kind = r.distance > vk.max_vision ? 0 : 2;
    }
    return r;
}

// Pass B, lane = screen row d_v, for screen column `col` (slot `k` of the wave's 64): the floor /
// ceiling cast (:102-126 / :135-153), then the wall column (:181-192), then the translucent
// overlays in ray order (:194-205), then the life bar (maze_discrete_3d.py:118-126).
// WALL says what the wave already knows about this 64-row chunk and the column's wall span (both wave-uniform):
//   0  the chunk misses the wall: no wall code at all;   1  it straddles an end of the wall (or the column has overlay
//   records): per-lane tests;   2  every row of the chunk is wall and the column has no overlay records: wall code only.
// The floor and the ceiling cast are ONE code path with per-lane selects (texture source, fog factor, translucency threshold,
// "paint outside the maze"): a chunk that holds ceiling AND floor rows — every chunk of a frame up to 64 rows high — used
// to run the cast twice, once per kind, each time with half its lanes.
template <int REC, bool STOCK, int WALL, class Flush>
__device__ __forceinline__ void pixel_pass(const ViewK &vk, const Task &t, double pos_x, double pos_y, const RowK &rk,
                                           const uint8_t *texts, const double *transp, const ColRecD &wc,
                                           const uint32_t *entries, int k, int d_v, double cs, double inv_cs,
                                           int cs_pow2, double text_to_cell, double inv_ttc, int ttc_pow2,
                                           int fast_tex, double tex_scale, int cell_shift, Flush &&flush, int &R, int &G, int &B) {
    const int n = t.n, TS = vk.TS;
    R = G = B = 0;
    bool tflag = false;
    const int span = wc.w_span;
    const bool in_wall = WALL == 0 ? false : (WALL == 2 ? true : (d_v >= (span & 0xfff) && d_v < ((span >> 12) & 0xfff)));
    const int n_tr = WALL == 2 ? 0 : (int)((unsigned)span >> 24);
    // ================= front half: every address and every fetch of this pixel ==========================================
    // The wall texel first (its address needs the column record and the row only), then the cast's: in a chunk that holds wall
    // AND floor / ceiling rows the two fetches travel together (round 5: they used to follow each other, and a third global
    // load — see `ceil_delta` — stood in front of the cast's).
    uint32_t wall_tx = 0;
    if (WALL != 0 && in_wall) {                                               // :181-188
        const double local_v = rk.ys * wc.w_ratio + t.agent_h;
        double d_j = (STOCK || vk.text_size_pow2) ? local_v * vk.inv_text_size : local_v / vk.text_size;
        d_j -= floor(d_j);
        wall_tx = vk.tex[(uint32_t)(wc.w_tex + (int)(TS * d_j))];
    }
    // A wall pixel overwrites whatever the floor / ceiling cast painted; the cast's only surviving
    // side effect is the transparent_array flag, which is read by the overlays alone. So the cast
    // can be skipped for wall pixels of columns without overlay records (bit-identical).
    const bool cast = WALL != 2 && rk.kind != 0 && !(WALL == 1 && in_wall && n_tr == 0);
    const bool fl = rk.kind == 1;                                             // floor :95-126, else ceiling :129-153
    bool paint = false, inside = false;
    uint32_t cast_tx = 0;
    double a = 0.0, tr = 0.0;
    if (cast) {
        const double eff = div_by(rk.distance, wc.cos_hp, wc.rcos_hp);
        // fog a = clamp(2*eff/max_vision - 1, 0, 1): when 2*eff is clearly below max_vision the
        // rounded quotient cannot exceed 1, so a == 0 without performing the division
        if (2.0 * eff > vk.max_vision_lo) {
            a = div_by(2.0 * eff, vk.max_vision, vk.inv_max_vision) - 1.0;
            a = a > 0.0 ? a : 0.0;
            a = a < 1.0 ? a : 1.0;
        }
        const double hit_x = eff * wc.cos_abs + pos_x;
        const double hit_y = eff * wc.sin_abs + pos_y;
        // Cell (i, j) and texel (ti, tj) of the hit point. When cell size, texture size and texture
        // resolution are powers of two (the stock task: 2.0, 1.0, 64) every scaling below is exact and
        // frac(frac(x / cs) * cs / ts) == frac(x / ts), so for x >= 0
        //     floor(x * TS / ts) = (cell index << cell_shift) | (texel index)
        // and one multiply + one convert + two integer ops per axis replace the chain of
        // divide / floor / subtract / scale steps (bit-identical; the general chain stays for other sizes).
        int i, j, ti, tj;
        if (__builtin_expect(fast_tex && hit_x >= 0.0 && hit_y >= 0.0, 1)) {
            const int xi = cell_index(hit_x * tex_scale), xj = cell_index(hit_y * tex_scale);
            i = xi >> cell_shift;
            j = xj >> cell_shift;
            ti = xi & (TS - 1);
            tj = xj & (TS - 1);
        } else {
            const double fi = cs_pow2 ? hit_x * inv_cs : hit_x / cs;
            const double fj = cs_pow2 ? hit_y * inv_cs : hit_y / cs;
            i = cell_index(fi);
            j = cell_index(fj);
            double d_i, d_j;
            if (fl) {                                                         // floor :107-116
                d_i = fi - floor(fi);
                d_j = fj - floor(fj);
                d_i = ttc_pow2 ? d_i * inv_ttc : d_i / text_to_cell;
                d_j = ttc_pow2 ? d_j * inv_ttc : d_j / text_to_cell;
                d_i -= floor(d_i);
                d_j -= floor(d_j);
            } else {                                                          // ceiling :139-143
                const double gi = (STOCK || vk.text_size_pow2) ? hit_x * vk.inv_text_size : hit_x / vk.text_size;
                const double gj = (STOCK || vk.text_size_pow2) ? hit_y * vk.inv_text_size : hit_y / vk.text_size;
                d_i = gi - floor(gi);
                d_j = gj - floor(gj);
            }
            ti = cell_index(d_i * TS);
            tj = cell_index(d_j * TS);
        }
        inside = (unsigned)i < (unsigned)n && (unsigned)j < (unsigned)n;
        // the floor is painted inside the maze only (:117), the ceiling everywhere (:144-146)
        paint = inside || !fl;
        if (paint) {
            // 24-bit multiplies are full rate (v_mad_u32_u24); cells and texture rows are far below 2^24
            const uint32_t cell = inside ? __umul24(i, n) + j : 0u;
#ifdef MG_MAZE3D_KNOCKOUT_TEXADDR     /* timing experiment only: every lane fetches from one 256-byte stretch of the texture */
            const uint32_t row = (uint32_t)tj & 63u;
#else
            const uint32_t row = __umul24((uint32_t)ti, TS) + (uint32_t)tj;
#endif
            // (a select between the two kernarg POINTERS compiled into a select of their kernarg ADDRESSES and a vector load of the
            // winner: a dependent global load in front of every texel fetch, at every frame size. One base, two offsets.)
            const long long toff = fl ? (long long)(__umul24(__umul24((uint32_t)texts[cell], TS), TS) + row) : vk.ceil_delta + (long long)row;
            cast_tx = vk.tex[toff];
            tr = transp[cell];
        }
    }
    // ================= the PREVIOUS pixel's store goes out here ===========================================================
    // gfx9-family vector memory returns in order and loads and stores share one counter (vmcnt): a wait for a texel fetched AFTER
    // the previous chunk's frame store is also a wait for that store's acknowledgement from L2 — every chunk of every wave paid the
    // write latency before its arithmetic could start (64 x 64 frames: stores alone 0.43 ms, everything but the stores ~0.5 ms, both
    // 0.91 ms: no overlap at all). Issued here — after this pixel's fetches, before their first use — the store is the YOUNGEST
    // operation in flight, the fetches are waited for with vmcnt(1), and the store's latency hides behind the colour arithmetic.
    flush();
    // ================= back half: colours =================================================================================
    if (cast && paint) {
        const uint32_t tx = cast_tx;
        const double alpha = fl ? a * rk.light : a;                           // :119 / :147
        const double oma = 1.0 - alpha;
#ifdef MG_MAZE3D_KNOCKOUT_COLOR      /* timing experiment only: no colour arithmetic */
        R = (int)(tx & 255u) + (int)oma; G = (int)((tx >> 8) & 255u); B = (int)(tx >> 16);
#else
        R = (int)(rk.light * (oma * tex_r(tx)));
        G = (int)(rk.light * (oma * tex_g(tx)));
        B = (int)(rk.light * (oma * tex_b(tx)));
#endif
        if (__builtin_expect(inside && tr > (fl ? 0.01 : 0.0), 0)) {          // :121-126 (> 0.01) / :148-153 (> 0)
            const double tf = tr * 0.50 + 0.10, om = 1.0 - tf;
            R = (int)(om * (double)R);
            G = (int)(om * (double)G + tf * 255.0);
            B = (int)(om * (double)B);
            tflag = true;
        }
    }
    if (WALL != 0 && in_wall) {                                               // :189-192
        const uint32_t tx = wall_tx;
        const double oma = wc.w_oma, light = wc.w_light;
        R = (int)(light * (oma * tex_r(tx)));
        G = (int)(light * (oma * tex_g(tx)));
        B = (int)(light * (oma * tex_b(tx)));
    }
    for (int q = 0; q < n_tr; ++q) {                                          // :194-205
        int lo, hi, cell;
        if (REC == 1) {
            const uint32_t en = entries[q * vk.slab + k];
            lo = (int)(en & 0xfffu); hi = (int)((en >> 12) & 0xfffu); cell = (int)(en >> 24);
        } else {
            const uint32_t e0 = entries[(q * vk.slab + k) * 2], e1 = entries[(q * vk.slab + k) * 2 + 1];
            lo = (int)(e0 & 0xffffu); hi = (int)(e0 >> 16); cell = (int)e1;
        }
        if (!tflag && d_v >= lo && d_v < hi) {
            const double tf = transp[cell] * 0.50 + 0.10, om = 1.0 - tf;
            R = (int)(om * (double)R);
            G = (int)(om * (double)G + tf * 255.0);
            B = (int)(om * (double)B);
        }
    }
}

__device__ __forceinline__ void py_slice(long a, long b, long len, int &lo, int &hi) {
    if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;
    if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
    lo = (int)a;
    hi = (int)b;
}

typedef int mz_v3i __attribute__((ext_vector_type(3)));   // 12-byte pixel, stored with one buffer_store_dwordx3

// REC: words per translucent-cell record, 1 (compact) or 2. STOCK: the stock renderer configuration, decided on the host
// (mg_maze3d_step) — every task's cell size the same power of two, texture size and resolution powers of two, int32 frames: the
// run-time `is this a power of two` flags of the general kernel become constants, and with them go the correctly-rounded
// divisions nobody takes, their scalar branches, the byte-output path and a quarter-rate 32-bit multiply in the store address.
// SMALL: the one-wave-per-env instantiation (frames up to 64 x 64): a column is ONE 64-row chunk there, so what the general kernel
// does once per column — the record's broadcast, 13 v_readlane + 4 converts — is paid per chunk, and at 64 x 64 the general kernel
// is VALU-bound (82 % busy). Pass A parks the widened record in LDS (ColRecLds) and the pixel pass reads it back with four
// broadcast ds_read_b128; five waves per SIMD give the record's VGPR copy its registers (the cap is not what bounds the resident
// waves: 4 / 5 / 6 measured alike). 64 x 64 x 65 536 envs: 0.913 -> 0.865 ms (profiles/r05/maze3d_small_frames.txt).
// U8 (stock instantiations only): uint8 frames, known at compile time — the general kernel reads vk.obs_u8 at run time. Up to
// round 5 uint8 frames always ran the GENERAL kernel, which is why the mode was no faster than int32 (its extra arithmetic cost
// what the smaller frames saved).
template <int REC, bool STOCK, bool SMALL = false, bool U8 = false>
__global__ __launch_bounds__(MZ_BLOCK) __attribute__((amdgpu_waves_per_eu(SMALL ? 5 : 6))) void maze3d_step_kernel(mg_maze_tasks T, mg_maze_state st, ViewK vk,
                                                               int task_type, int max_steps, int continuous,
                                                               int pre_moved, int auto_reset, int n_envs,
                                                               const void *action,
                                                               void *obs, float *reward, double *reward64,
                                                               uint8_t *done) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int e = mg::env_of_block(blockIdx.x, n_envs);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_threads = blockDim.x, n_waves = n_threads >> 6;   // 1, 2 or 4 waves per env (host picks by frame size)
    const Task t = load_task(T, st.task_id[e]);
    const int nn = t.nn;

    // ---- LDS carve-up ---------------------------------------------------------------------------
    EnvShared *es = reinterpret_cast<EnvShared *>(smem);
    size_t off = (sizeof(EnvShared) + 15) & ~size_t(15);
    double *transp = reinterpret_cast<double *>(smem + off);
    off += sizeof(double) * nn;
    uint32_t *entries_all = reinterpret_cast<uint32_t *>(smem + off);
    off += sizeof(uint32_t) * REC * vk.slab * vk.t_max * n_waves;
    int8_t *walls = reinterpret_cast<int8_t *>(smem + off);
    off += (nn + 15) & ~15;
    uint8_t *texts = reinterpret_cast<uint8_t *>(smem + off);
    off += (nn + 15) & ~15;
    off = (off + 15) & ~size_t(15);
    RowRec *row_tab = reinterpret_cast<RowRec *>(smem + off);      // [V rounded up to 64]: distance, light, ys, kind of a screen row
    const int v_pad = (vk.V + 63) & ~63;
    ColRecLds *colrec = reinterpret_cast<ColRecLds *>(smem + off + sizeof(RowRec) * (size_t)v_pad);   // SMALL only: [slab]

    // ---- phase 0: transition + scalar part of evaluation_rule (one thread) ----------------------
    if (tid == 0) {
        Agent a = load_agent(st, n_envs, e);
        es->done = 0;
        es->reward = 0.0;
        if (action != nullptr) {
            if (continuous) {
                if (!pre_moved) {   // otherwise maze_cont_move_kernel already advanced every env
                    const float *ac = static_cast<const float *>(action) + 2 * (size_t)e;
                    transition_continuous(t, vk.col_dist, ac[0], ac[1], a);
                }
            } else {
                transition_discrete(t, static_cast<const int32_t *>(action)[e] & 3, a);
            }
            double r;
            es->done = eval_scalar(t, st, e, task_type, max_steps, a, r);
            es->reward = r;
            if (reward) reward[e] = (float)r;
            if (reward64) reward64[e] = r;
            done[e] = (uint8_t)es->done;
        }
        es->a = a;
    }
    __syncthreads();
    if (action != nullptr && task_type == MG_MAZE_SURVIVAL) eval_cells(t, st, e, tid, n_threads);   // :83-88
    if (es->done && auto_reset) {
        if (task_type == MG_MAZE_SURVIVAL) reset_cells(t, st, e, tid, n_threads, false);
        __syncthreads();
        if (tid == 0) reset_agent(t, task_type, es->a);
    }
    __syncthreads();
    if (tid == 0) {
        Agent &a = es->a;
        if (action != nullptr) store_agent(st, n_envs, e, a);
        if (continuous) {
            es->pos[0] = (double)a.lx;
            es->pos[1] = (double)a.ly;
            es->s_ori = sin(a.ori);
            es->c_ori = cos(a.ori);
        } else {
            es->pos[0] = a.gx * t.cell_size + 0.5 * t.cell_size;   // get_cell_center
            es->pos[1] = a.gy * t.cell_size + 0.5 * t.cell_size;
            es->s_ori = (double)vk.ori_sin[a.ori_idx];
            es->c_ori = (double)vk.ori_cos[a.ori_idx];
        }
    }
    // ---- phase 1: stage the task grids and the per-row constants in LDS ----------------------------
    for (int r = tid; r < v_pad; r += n_threads) {
        RowK rk{0, 0.0, 0.0, 0.0};
        if (r < vk.V) rk = row_constants(vk, t, r);
        row_tab[r] = RowRec{rk.distance, rk.light, rk.ys, rk.kind, 0};
    }
    for (int c = tid; c < nn; c += n_threads) {
        walls[c] = t.walls[c];
        texts[c] = t.texts[c];
        if (task_type == MG_MAZE_SURVIVAL) transp[c] = st.cur_food[fidx(st, e, c)];       // alias maze_base.py:57
        else transp[c] = (c == t.gx * t.n + t.gy) ? 1.0 : 0.0;                            // :59-60
    }
    __syncthreads();

    // ---- phase 2: render ---------------------------------------------------------------------------
    const double cs = t.cell_size;
    int ex;
    const int cs_pow2 = STOCK ? 1 : (frexp(cs, &ex) == 0.5);
    const double inv_cs = 1.0 / cs;
    const double text_to_cell = vk.text_size / cs;
    const int ttc_pow2 = STOCK ? 1 : (frexp(text_to_cell, &ex) == 0.5);
    const double inv_ttc = 1.0 / text_to_cell;
    // integer texel / cell addressing (see pixel_pass): all three sizes powers of two, cells at least one
    // texture wide, and every reachable coordinate times TS / text_size far below 2^31
    const double tex_scale = (double)vk.TS * vk.inv_text_size;
    const int ts_pow2 = (vk.TS & (vk.TS - 1)) == 0;
    const int fast_tex = STOCK ? 1 : (cs_pow2 && ttc_pow2 && vk.text_size_pow2 && ts_pow2 && inv_ttc >= 1.0 &&
                                      (vk.eff_max + cs * (double)t.n) * tex_scale < 1073741824.0);
    const int cell_shift = fast_tex ? ilogb(inv_ttc * (double)vk.TS) : 0;

    int lb_x0 = 0, lb_x1 = 0, lb_y0 = 0, lb_y1 = 0;   // life bar rectangle, maze_discrete_3d.py:118-126
    if (task_type == MG_MAZE_SURVIVAL) {
        const double lifebar_l = es->a.life / t.max_life * (0.80 * vk.V);
        const double sx = 0.10 * vk.V, sy = 0.10 * vk.V;
        py_slice((long)sx, (long)(sx + lifebar_l), vk.H, lb_x0, lb_x1);
        py_slice((long)sy, (long)(sy + 0.05 * vk.H), vk.V, lb_y0, lb_y1);
        // the rectangle is the env's, i.e. wave-uniform, but it was computed from an LDS value: say so, and the per-column /
        // per-chunk "does the bar cross here" tests below become scalar branches instead of five VALU ops on every pixel
        lb_x0 = __builtin_amdgcn_readfirstlane(lb_x0); lb_x1 = __builtin_amdgcn_readfirstlane(lb_x1);
        lb_y0 = __builtin_amdgcn_readfirstlane(lb_y0); lb_y1 = __builtin_amdgcn_readfirstlane(lb_y1);
    }

    uint32_t *entries = entries_all + (size_t)wave * vk.slab * vk.t_max * REC;
    const double pos_x = es->pos[0], pos_y = es->pos[1];   // registers: the pixel loop must not re-read LDS for them
    int32_t *img = static_cast<int32_t *>(obs) + (size_t)e * vk.H * vk.V * 3;
    uint8_t *img8 = static_cast<uint8_t *>(obs) + (size_t)e * vk.H * vk.V * 3;
    // columns are dealt to the waves in equal slabs (<= vk.slab each) so narrow images keep all waves busy
    const int slab = min(vk.slab, (vk.H + n_waves - 1) / n_waves);
    // With four waves per env (large frames) the columns of a group of 4 * slab are dealt round-robin — wave w takes columns
    // w, w + 4, ... — so the waves of an env write ADJACENT columns at about the same time: one contiguous stretch of the frame per
    // env instead of four streams 32 columns (98 KB at 256 x 256) apart. The frame stores alone (everything else knocked out) run
    // 6.5 % faster that way at 256 x 256 (2.37 -> 2.22 ms, profiles/r04/maze3d_store_pattern.txt); small frames (one or two waves
    // per env) keep consecutive columns per wave, which measured faster there.
    const int col_step = n_waves >= 4 ? n_waves : 1;
    for (int gbase = 0; gbase < vk.H; gbase += n_waves * slab) {
        const int cbase = col_step == 1 ? gbase + wave * slab : gbase + wave;
        const int ncols = col_step == 1 ? min(slab, vk.H - cbase) : min(slab, (vk.H - cbase + n_waves - 1) / n_waves);   // (<= 0: nothing left)
        ColRec mine{};
        if (lane < ncols)
            mine = column_pass<REC, STOCK>(vk, t, *es, walls, texts, transp, cbase + lane * col_step, lane, entries, cs, inv_cs, cs_pow2);
        if constexpr (SMALL) {                    // widened once per lane = per column, parked in LDS for the pixel pass
            // the host reserves sizeof(ColRecLds) * vk.slab for colrec (mg_maze3d_step's `lds`), and ncols <= slab <= vk.slab:
            // only lanes that own a column may store, lanes ncols..63 would write past the dynamic LDS allocation
            const ColRecD d = widen(mine);
            if (lane < ncols) colrec[lane] = ColRecLds{d.cos_hp, d.cos_abs, d.sin_abs, d.rcos_hp, d.w_oma, d.w_ratio, d.w_light, d.w_tex, d.w_span};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // column-major walk: column k's record is broadcast ONCE (v_readlane + converts; SMALL: read back from LDS) and then
        // reused by all V/64 row chunks; the per-row constants come from the LDS row table
        const bool obs_u8 = STOCK ? U8 : (vk.obs_u8 != 0);
        const uint32_t px_bytes = obs_u8 ? 3u : 12u;
#ifdef MG_MAZE3D_KNOCKOUT_PIXELS      /* timing experiment only: everything but the pixel loop */
        if (mine.w_span == 0x7fffffff) ((int *)obs)[e] = 1;
        continue;
#endif
        const RowRec *row_lane = row_tab + lane;                  // this lane's row of every 64-row chunk: one add per chunk away
        const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc(
            obs_u8 ? static_cast<void *>(img8) : static_cast<void *>(img), 0, (int)((uint32_t)vk.H * (uint32_t)vk.V * px_bytes), 0x00020000);
        // The pixel computed by one chunk iteration is stored DURING THE NEXT one (see pixel_pass: after that pixel's texel fetches
        // went out, before they are used), so a fetch never waits behind the frame store in the in-order vmcnt queue. The store is
        // a BUFFER store with the env's frame as its range (streaming, `nt`: a 12.9 GB frame batch can never stay in the 32 MB of
        // L2, but the textures and task tables it would evict are re-read by every pixel): a lane without a pending pixel — the
        // first chunk; rows past V in a ragged last chunk — sends an out-of-range offset and the hardware drops its write. No
        // branch around the store: the wait-count pass can then prove the store is in flight when the fetches are awaited and emits
        // vmcnt(1); behind an `if (pending)` it has to assume the store may be missing and waits with vmcnt(0), store included.
        // Worth 2.5 % at 64 x 64 and 0.7 % (discrete) / 1.7 % (continuous) at 256 x 256 (profiles/r05/maze3d_small_frames.txt).
        int p_r = 0, p_g = 0, p_b = 0;
        uint32_t p_off = 0;
        bool p_ok = false;
        // uint8 frames (non-parity fast path: saturate to a byte): three byte stores per pixel. Round 6 also built the PACKED store
        // the round-5 review asked for — a chunk of a column is 64 pixels x 3 bytes = 192 contiguous bytes; each lane packs its pixel
        // into 24 bits, three DPP quad broadcasts bring a quad's four words to every lane, lane 4k stores the quad's 12 bytes as one
        // buffer_store_dwordx3 (16 store requests per wave instead of 192; needs res_v % 4 == 0) — and measured it SLOWER: 262
        // instead of 778 store instructions per wave, but 21 595 instead of 18 512 VALU instructions, 2.524 against 2.415 ms at
        // 256 x 256 x 16 384 envs (int32 frames: 2.515 ms) and 0.956 against 0.922 ms at 64 x 64 x 65 536. With the 4x smaller frames
        // the store stream no longer limits the kernel, VALU issue does (floors 1.97 / 2.30 ms for the two variants):
        // profiles/r06/maze3d_uint8.txt. The packed path stays behind MG_MAZE3D_U8_PACKED=1 so that the measurement can be repeated;
        // byte stores are the default.
        const bool u8_packed = obs_u8 && (vk.V & 3) == 0 && vk.obs_u8 == 2;
        const bool quad_lead = (lane & 3) == 0;
        auto flush = [&]() {
            const uint32_t o = p_ok ? p_off : 0x80000000u;      // (frames are < 2^31 bytes: res_h <= 32767, res_v <= 4095)
            if (u8_packed) {
                const uint32_t w0 = (uint32_t)min(max(p_r, 0), 255) | ((uint32_t)min(max(p_g, 0), 255) << 8) | ((uint32_t)min(max(p_b, 0), 255) << 16);
                const uint32_t w1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)w0, 0x55, 0xf, 0xf, true);      // quad_perm [1,1,1,1]
                const uint32_t w2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)w0, 0xAA, 0xf, 0xf, true);      // quad_perm [2,2,2,2]
                const uint32_t w3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)w0, 0xFF, 0xf, 0xf, true);      // quad_perm [3,3,3,3]
                mz_v3i v;          // bytes r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3 (only lane 4k's own w0 is pixel 4k's)
                v.x = (int)(w0 | (w1 << 24)); v.y = (int)((w1 >> 8) | (w2 << 16)); v.z = (int)((w2 >> 16) | (w3 << 8));
                __builtin_amdgcn_raw_buffer_store_b96(v, frame, quad_lead ? o : 0x80000000u, 0, 2);
            } else if (obs_u8) {
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)min(max(p_r, 0), 255), frame, o, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)min(max(p_g, 0), 255), frame, o + 1u, 0, 2);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)min(max(p_b, 0), 255), frame, o + 2u, 0, 2);
            } else {
                mz_v3i v;
                v.x = p_r; v.y = p_g; v.z = p_b;
                __builtin_amdgcn_raw_buffer_store_b96(v, frame, o, 0, 2);       // aux 2 = nt
            }
        };
        for (int k = 0; k < ncols; ++k) {
            ColRecD wc;
            if constexpr (SMALL) {
                const ColRecLds r = colrec[k];                       // uniform address: LDS broadcast reads
                wc = ColRecD{r.cos_hp, r.cos_abs, r.sin_abs, r.rcos_hp, r.w_oma, r.w_ratio, r.w_light, r.w_tex,
                             __builtin_amdgcn_readfirstlane(r.w_span)};
            } else wc = widen(bcast(mine, k));
            const int col = __builtin_amdgcn_readfirstlane(cbase + k * col_step);
            const bool in_lb_x = col >= lb_x0 && col < lb_x1;
            // frame-relative byte offset of this lane's pixel in chunk 0 of the column (< 4 GiB); a chunk further down is a
            // scalar away — no per-pixel multiply (a 32-bit one is quarter rate)
            const uint32_t px_index = (uint32_t)(col * vk.V) + (uint32_t)lane;
            // (x * 12 as two full-rate shift-adds: v_mul_lo_u32 is quarter rate, and at one chunk per column this is per-chunk work)
            uint32_t px3;
            asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(px3) : "v"(px_index));       // (spelled out: the optimiser folds it back into the multiply)
            uint32_t pix_lane = obs_u8 ? px3 : px3 << 2;
            asm volatile("" : "+v"(pix_lane));      // (keep the product: folded into the chunk offset it is multiplied again per pixel)
            uint32_t chunk_bytes = 0;                 // rbase * px_bytes, carried as a scalar of its own (an add per chunk)
            for (int rbase = 0; rbase < vk.V; rbase += 64, chunk_bytes += 64u * px_bytes) {
                asm volatile("" : "+s"(chunk_bytes));
                const int d_v = rbase + lane;
                const bool row_ok = d_v < vk.V;
                const RowRec rr = row_lane[rbase];
                RowK rk;
                rk.kind = rr.kind;
                rk.distance = rr.distance;
                rk.light = rr.light;
                rk.ys = rr.ys;
                int R, G, B;
                // where the chunk lies relative to the column's wall span [v_s, v_e) — scalars, one branch per chunk
                const int w_s = wc.w_span & 0xfff, w_e = (wc.w_span >> 12) & 0xfff;
#define MG_PIXEL(WALL_)                                                                                                      \
    pixel_pass<REC, STOCK, WALL_>(vk, t, pos_x, pos_y, rk, texts, transp, wc, entries, k, d_v, cs, inv_cs, cs_pow2, text_to_cell, \
                                  inv_ttc, ttc_pow2, fast_tex, tex_scale, cell_shift, flush, R, G, B)
#ifdef MG_MAZE3D_KNOCKOUT_COMPUTE     /* timing experiment only: the frame stores alone, in the kernel's own pattern */
                R = d_v + w_s; G = col; B = rk.kind;
                flush();
#else
                if ((unsigned)wc.w_span < 0x1000000u && rbase >= w_s && rbase + 64 <= w_e) MG_PIXEL(2);
                else if ((unsigned)wc.w_span < 0x1000000u && (rbase + 64 <= w_s || rbase >= w_e)) MG_PIXEL(0);
                else MG_PIXEL(1);
#endif
#undef MG_PIXEL
                if (in_lb_x && rbase < lb_y1 && rbase + 64 > lb_y0) {          // wave-uniform: the bar crosses this chunk
                    if (d_v >= lb_y0 && d_v < lb_y1) { R = 255; G = 0; B = 0; }
                }
                p_r = R; p_g = G; p_b = B; p_off = pix_lane + chunk_bytes; p_ok = row_ok;
            }
        }
        flush();            // the last pixel of this group of columns
        p_ok = false;
        __builtin_amdgcn_wave_barrier();
    }
}

// The continuous transition is ~10 k dependent scalar ops per env (10 sub-steps x 9 neighbour cells
// of soft collision, dynamics.py:71-92). Inside the render kernel it would run on one thread while
// 255 wait; as its own lane-per-env launch all envs advance in parallel.
__global__ __launch_bounds__(MZ_BLOCK) void maze_cont_move_kernel(mg_maze_tasks T, mg_maze_state st, double col_dist,
                                                                  int n_envs, const float *action) {
    const int e = blockIdx.x * MZ_BLOCK + threadIdx.x;
    if (e >= n_envs) return;
    const Task t = load_task(T, st.task_id[e]);
    Agent a = load_agent(st, n_envs, e);
    transition_continuous(t, col_dist, action[2 * (size_t)e], action[2 * (size_t)e + 1], a);
    st.grid[e] = a.gx;
    st.grid[n_envs + e] = a.gy;
    st.ori[e] = a.ori;
    st.loc[e] = a.lx;
    st.loc[n_envs + e] = a.ly;
}

__global__ __launch_bounds__(MZ_BLOCK) void maze_reset_kernel(mg_maze_tasks T, mg_maze_state st, int task_type,
                                                              int n_envs, const uint8_t *mask) {
    // one wave per env so the SURVIVAL cell arrays are written coalesced
    const int e = (blockIdx.x * MZ_BLOCK + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (e >= n_envs) return;
    if (mask != nullptr && mask[e] == 0) return;
    const Task t = load_task(T, st.task_id[e]);
    if (task_type == MG_MAZE_SURVIVAL) reset_cells(t, st, e, lane, 64, true);
    if (lane == 0) {
        Agent a = load_agent(st, n_envs, e);
        reset_agent(t, task_type, a);
        store_agent(st, n_envs, e, a);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

int check_tasks(const mg_maze_tasks *T, int task_type) {
    if (T->n < 3 || T->n > 255 || T->n_tasks < 1) return mg::set_error(MG_ERR_BAD_SIZE, "maze n=%d n_tasks=%d", T->n, T->n_tasks);
    if (!T->start || !T->goal || !T->walls || !T->texts || !T->scalars)
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_maze_tasks has a NULL array");
    if (task_type == MG_MAZE_SURVIVAL && (!T->food_rewards || !T->food_interval))
        return mg::set_error(MG_ERR_NULL_POINTER, "SURVIVAL needs food_rewards and food_interval");
    if (task_type != MG_MAZE_SURVIVAL && task_type != MG_MAZE_ESCAPE)
        return mg::set_error(MG_ERR_BAD_CONFIG, "task_type %d", task_type);
    return MG_OK;
}

int check_mstate(const mg_maze_state *s, int task_type) {
    if (!s->task_id || !s->grid || !s->steps) return mg::set_error(MG_ERR_NULL_POINTER, "mg_maze_state has a NULL array");
    if (task_type == MG_MAZE_SURVIVAL && (!s->life || !s->cur_food || !s->wait_refresh || !s->revival))
        return mg::set_error(MG_ERR_NULL_POINTER, "SURVIVAL needs life / cur_food / wait_refresh / revival");
    return MG_OK;
}

int check_slots(const mg_maze_tasks *T, const mg_maze_state *s, int task_type) {
    if (task_type == MG_MAZE_SURVIVAL && s->food_by_slot &&
        (!T->food_cells || !T->n_food || !T->cell_slot || !T->slot_food || !T->slot_interval || T->max_food < 1))
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_maze_state.food_by_slot needs mg_maze_tasks.food_cells / n_food / cell_slot / "
                             "slot_food / slot_interval");
    return MG_OK;
}

}  // namespace

extern "C" int mg_maze_view_tables(int32_t res_h, double tan_half_fov, double l_focal, double *col_cos,
                                   double *col_sin) {
    MG_REQUIRE_PTR(col_cos);
    MG_REQUIRE_PTR(col_sin);
    if (res_h <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "res_h=%d", res_h);
    // ray_caster_utils.py:68-90, host doubles (sqrt and / are correctly rounded like numpy's)
    const double half_h = tan_half_fov * l_focal;
    const double pixel_size = 2.0 * half_h / res_h;
    const double pixel_factor = pixel_size / l_focal;
    double tan_hp = (-0.5 - res_h / 2.0) * pixel_factor;
    for (int d_h = 0; d_h < res_h; ++d_h) {
        tan_hp += pixel_factor;
        const double chp = sqrt(1.0 / (1.0 + tan_hp * tan_hp));
        col_cos[d_h] = chp;
        col_sin[d_h] = tan_hp * chp;
    }
    return MG_OK;
}

extern "C" int mg_maze_reset(const mg_maze_tasks *T, int32_t task_type, int32_t n, const mg_maze_state *st,
                             const uint8_t *mask, void *stream) {
    MG_REQUIRE_PTR(T);
    MG_REQUIRE_PTR(st);
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (int rc = check_tasks(T, task_type)) return rc;
    if (int rc = check_mstate(st, task_type)) return rc;
    if (int rc = check_slots(T, st, task_type)) return rc;
    const long threads = (long)n * 64;
    mg::DeviceGuard guard(mg::device_of(st->grid));
    hipLaunchKernelGGL(maze_reset_kernel, dim3((unsigned)((threads + MZ_BLOCK - 1) / MZ_BLOCK)), dim3(MZ_BLOCK), 0,
                       (hipStream_t)stream, *T, *st, task_type, n, mask);
    return mg::check_launch("maze_reset_kernel");
}

extern "C" int mg_maze2d_step(const mg_maze_tasks *T, int32_t task_type, int32_t max_steps, int32_t view_grid,
                              int32_t auto_reset, int32_t n, const mg_maze_state *st, const int32_t *action,
                              float *obs, float *reward, double *reward64, uint8_t *done, void *stream) {
    MG_REQUIRE_PTR(T);
    MG_REQUIRE_PTR(st);
    MG_REQUIRE_PTR(obs);
    if (action != nullptr) MG_REQUIRE_PTR(done);
    if (n <= 0 || view_grid < 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d view_grid=%d", n, view_grid);
    if (int rc = check_tasks(T, task_type)) return rc;
    if (int rc = check_mstate(st, task_type)) return rc;
    if (int rc = check_slots(T, st, task_type)) return rc;
    mg::DeviceGuard guard(mg::device_of(st->grid));
    hipLaunchKernelGGL(maze2d_step_kernel, dim3((n + MZ_BLOCK - 1) / MZ_BLOCK), dim3(MZ_BLOCK), 0, (hipStream_t)stream,
                       *T, *st, task_type, max_steps, view_grid, auto_reset, n, action, obs, reward, reward64, done);
    return mg::check_launch("maze2d_step_kernel");
}

// ---- mg_maze_view.uniform_cell_size is a promise about the task table; a wrong one would render wrong frames. The library
// checks it once per (scalars pointer, n_tasks, value): the [T][8] scalar rows are read back (T * 64 bytes) and every
// cell_size compared. Validated triples are remembered in a small process-wide table (mutex; 32 entries, oldest replaced).
namespace {
struct UniformEntry { const void *scalars; int n_tasks; double cs; };
std::mutex g_uniform_mu;
UniformEntry g_uniform[32];
int g_uniform_n = 0, g_uniform_next = 0;

bool uniform_cache_has(const void *scalars, int n_tasks, double cs) {
    std::lock_guard<std::mutex> lk(g_uniform_mu);
    for (int i = 0; i < g_uniform_n; ++i)
        if (g_uniform[i].scalars == scalars && g_uniform[i].n_tasks == n_tasks && g_uniform[i].cs == cs) return true;
    return false;
}
void uniform_cache_drop(const void *scalars) {
    std::lock_guard<std::mutex> lk(g_uniform_mu);
    for (int i = 0; i < g_uniform_n; ++i)
        if (g_uniform[i].scalars == scalars) g_uniform[i] = UniformEntry{nullptr, 0, 0.0};
}
void uniform_cache_put(const void *scalars, int n_tasks, double cs) {
    std::lock_guard<std::mutex> lk(g_uniform_mu);
    for (int i = 0; i < g_uniform_n; ++i)
        if (g_uniform[i].scalars == scalars) { g_uniform[i] = UniformEntry{scalars, n_tasks, cs}; return; }
    for (int i = 0; i < g_uniform_n; ++i)               // a slot emptied by uniform_cache_drop is reused before anything is evicted
        if (g_uniform[i].scalars == nullptr) { g_uniform[i] = UniformEntry{scalars, n_tasks, cs}; return; }
    const int slot = g_uniform_n < 32 ? g_uniform_n++ : (g_uniform_next++ & 31);
    g_uniform[slot] = UniformEntry{scalars, n_tasks, cs};
}
}  // namespace

extern "C" int mg_maze_forget_tasks(const mg_maze_tasks *T) {
    MG_REQUIRE_PTR(T);
    if (T->scalars) uniform_cache_drop(T->scalars);
    return MG_OK;
}

extern "C" int mg_maze_check_uniform_cell_size(const mg_maze_tasks *T, double uniform_cell_size, void *stream) {
    MG_REQUIRE_PTR(T);
    MG_REQUIRE_PTR(T->scalars);
    if (T->n_tasks <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_tasks = %d", T->n_tasks);
    if (!(uniform_cell_size > 0.0)) return mg::set_error(MG_ERR_BAD_CONFIG, "uniform_cell_size = %.17g is not a cell size", uniform_cell_size);
    uniform_cache_drop(T->scalars);                     // an explicit call always re-reads the table
    std::vector<double> rows((size_t)T->n_tasks * 8);
    const int device = mg::device_of(T->scalars);
    if (device >= 0) {
        mg::DeviceGuard guard(device);
        if (int rc = mg::check_hip(hipMemcpyAsync(rows.data(), T->scalars, rows.size() * sizeof(double), hipMemcpyDeviceToHost,
                                                  (hipStream_t)stream), "hipMemcpyAsync(task scalars)")) return rc;
        if (int rc = mg::check_hip(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize(task scalars)")) return rc;
    } else {
        memcpy(rows.data(), T->scalars, rows.size() * sizeof(double));      // a host table (CPU-side callers, tests)
    }
    for (int t = 0; t < T->n_tasks; ++t)
        if (rows[(size_t)t * 8] != uniform_cell_size)
            return mg::set_error(MG_ERR_BAD_CONFIG, "mg_maze_view.uniform_cell_size = %.17g but task %d of %d has cell_size %.17g: pass 0 "
                                 "(tasks differ) or the table's one cell size", uniform_cell_size, t, T->n_tasks, rows[(size_t)t * 8]);
    uniform_cache_put(T->scalars, T->n_tasks, uniform_cell_size);
    return MG_OK;
}

extern "C" int mg_maze3d_step(const mg_maze_tasks *T, const mg_maze_view *view, int32_t task_type, int32_t max_steps,
                              int32_t continuous, int32_t auto_reset, int32_t n, const mg_maze_state *st,
                              const void *action, void *obs, float *reward, double *reward64, uint8_t *done,
                              void *stream) {
    MG_REQUIRE_PTR(T);
    MG_REQUIRE_PTR(view);
    MG_REQUIRE_PTR(st);
    MG_REQUIRE_PTR(obs);
    if (action != nullptr) MG_REQUIRE_PTR(done);
    if (n <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "n_envs=%d", n);
    if (int rc = check_tasks(T, task_type)) return rc;
    if (int rc = check_mstate(st, task_type)) return rc;
    if (task_type == MG_MAZE_SURVIVAL && st->food_by_slot)
        return mg::set_error(MG_ERR_UNSUPPORTED, "mg_maze3d_step reads the SURVIVAL arrays by cell (food_by_slot is the 2-D kernel's layout)");
    if (continuous && (!st->ori || !st->loc)) return mg::set_error(MG_ERR_NULL_POINTER, "continuous needs ori / loc");
    if (!continuous && !st->ori_idx) return mg::set_error(MG_ERR_NULL_POINTER, "discrete needs ori_idx");
    if (view->res_h <= 0 || view->res_v <= 0 || view->res_v > 4095 || view->res_h > 32767)
        return mg::set_error(MG_ERR_BAD_SIZE, "resolution %d x %d", view->res_h, view->res_v);
    if (!view->col_cos || !view->col_sin || !view->textures || !view->ceil_texture)
        return mg::set_error(MG_ERR_NULL_POINTER, "mg_maze_view has a NULL table");
    if (view->tex_size <= 0 || view->n_textures <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "texture table");
    if (view->uniform_cell_size > 0.0) {
        // Before anything is launched: the promise is checked, not trusted: once per (table, value) — see mg_maze_check_uniform_cell_size.
        if (!uniform_cache_has(T->scalars, T->n_tasks, view->uniform_cell_size)) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
                return mg::set_error(MG_ERR_BAD_CONFIG, "mg_maze3d_step under stream capture with an unchecked mg_maze_view.uniform_cell_size "
                                     "= %.17g: call mg_maze_check_uniform_cell_size() on this task table before capturing",
                                     view->uniform_cell_size);
            (void)hipGetLastError();
            if (int rc = mg_maze_check_uniform_cell_size(T, view->uniform_cell_size, stream)) return rc;
        }
    }

    ViewK vk;
    vk.H = view->res_h;
    vk.V = view->res_v;
    vk.TS = view->tex_size;
    static const bool u8_pack = getenv("MG_MAZE3D_U8_PACKED") != nullptr;      // timing A/B only: the packed dwordx3 store (see flush())
    vk.obs_u8 = view->obs_format == 1 ? (u8_pack ? 2 : 1) : 0;
    vk.max_vision = view->max_vision;
    {
        // div_by()'s correctly-rounded quotient (Markstein) needs a divisor whose significand is not all ones;
        // cos_hp is float32-valued, so only a pathological max_vision could violate it
        uint64_t bits;
        memcpy(&bits, &view->max_vision, sizeof(bits));
        if ((bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull)
            return mg::set_error(MG_ERR_UNSUPPORTED, "max_vision = %.17g has an all-ones significand", view->max_vision);
    }
    vk.inv_max_vision = 1.0 / view->max_vision;
    vk.max_vision_lo = view->max_vision * (1.0 - 1.0e-12);   // see pixel_pass: below this, fog is exactly 0
    vk.l_focal = view->l_focal;
    vk.text_size = view->text_size;
    vk.inv_text_size = 1.0 / view->text_size;
    int ex;
    vk.text_size_pow2 = (frexp(view->text_size, &ex) == 0.5);
    vk.eff_max = view->max_vision * sqrt(1.0 + view->tan_half_fov * view->tan_half_fov) * 1.0001;
    vk.half_h = view->tan_half_fov * view->l_focal;              // ray_caster_utils.py:68
    vk.half_v = vk.half_h * vk.V / vk.H;                         // :69
    vk.pixel_size = 2.0 * vk.half_h / vk.H;                      // :70
    vk.col_dist = view->collision_dist;
    vk.col_cos = view->col_cos;
    vk.col_sin = view->col_sin;
    for (int i = 0; i < 4; ++i) { vk.ori_sin[i] = view->ori_sin[i]; vk.ori_cos[i] = view->ori_cos[i]; }
    vk.tex = view->textures;
    vk.ceil_tex = view->ceil_texture;
    vk.ceil_delta = (long long)(((intptr_t)view->ceil_texture - (intptr_t)view->textures) / (intptr_t)sizeof(uint32_t));
    // bound on translucent records per ray: one per DDA step plus the start cell. A ray advances
    // at least one cell per step and stops at max_vision or at the maze border.
    vk.t_max = 2 * T->n + 1;
    if (view->max_ray_records > 0 && view->max_ray_records < vk.t_max) vk.t_max = view->max_ray_records;
    // column_pass hands the record count to pixel_pass in 7 bits of the packed span word (span | n_tr << 24):
    // more than 127 translucent cells on one ray needs n > 63 AND a vision range spanning them; capped, documented
    // in metagym_hip.h (mg_maze_view.max_ray_records)
    if (vk.t_max > 127) vk.t_max = 127;
    // Waves per env (measured on MI355X, profiles/r04/maze3d_small_frames.txt; 256x256: profiles/r03/maze3d_waves_sweep.txt):
    // up to 64x64 one wave per env is fastest (32x32: 0.52 vs 0.60 ms with two at 65 536 envs, 64x64: 0.91 vs 0.95 — no
    // workgroup barrier partners, more independent envs resident per CU), 84x84 runs best with two, from 128x128 up with four.
    // 32-column slabs beat 64 everywhere (half the overlay-record LDS, more workgroups per CU).
    const long frame_px = (long)vk.H * vk.V;
    int n_waves = frame_px <= 64L * 64L ? 1 : (frame_px < 128L * 128L ? 2 : MZ_WAVES);
    vk.slab = SLAB;
    {
        // tuning override MG_MAZE3D_WAVES="<waves>[,<slab>]", read ONCE per process (thread-safe static
        // initialisation): a re-entrant ABI must not consult mutable process state on every step
        struct Override { int waves = 0, slab = 0; };
        static const Override ov = [] {
            Override o;
            if (const char *ev = getenv("MG_MAZE3D_WAVES")) {
                int w = 0, sl = 0;
                if (sscanf(ev, "%d,%d", &w, &sl) >= 1 && (w == 1 || w == 2 || w == 4)) {
                    o.waves = w;
                    o.slab = (sl == 32 || sl == 64) ? sl : SLAB;
                }
            }
            return o;
        }();
        if (ov.waves) { n_waves = ov.waves; vk.slab = ov.slab; }
    }
    const int rec = (vk.V < 4096 && T->n * T->n <= 256) ? 1 : 2;
    static const bool no_small = getenv("MG_MAZE3D_NO_SMALL") != nullptr;
    const bool small = n_waves == 1 && !no_small;       // one wave per env: the SMALL instantiation (column records parked in LDS)
    const size_t lds = ((sizeof(EnvShared) + 15) & ~size_t(15)) + sizeof(double) * T->n * T->n +
                       sizeof(uint32_t) * rec * vk.slab * vk.t_max * n_waves +
                       2 * ((size_t)(T->n * T->n + 15) & ~size_t(15)) + sizeof(RowRec) * (size_t)((vk.V + 63) & ~63) + 32 +
                       (small ? sizeof(ColRecLds) * (size_t)vk.slab : 0);
    if (lds > 160 * 1024) return mg::set_error(MG_ERR_BAD_SIZE, "maze n=%d needs %zu B of LDS (> 160 KiB)", T->n, lds);
    const int device = mg::device_of(obs);
    mg::DeviceGuard guard(device);
    if (lds > 64 * 1024) {
        // the opt-in is sticky per device: raise it when a larger request comes, not on every step
        static std::atomic<size_t> granted[MG_MAX_DEVICES];
        const int slot = (device >= 0 && device < MG_MAX_DEVICES) ? device : 0;
        if (device < 0 || device >= MG_MAX_DEVICES || lds > granted[slot].load(std::memory_order_relaxed)) {
            for (const void *fn : {reinterpret_cast<const void *>(maze3d_step_kernel<1, false>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<2, false>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<1, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<2, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<1, false, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<2, false, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<1, true, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<2, true, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<1, true, false, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<2, true, false, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<1, true, true, true>),
                                   reinterpret_cast<const void *>(maze3d_step_kernel<2, true, true, true>)}) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return mg::check_hip(e, "hipFuncSetAttribute(maze3d_step_kernel)");
            }
            granted[slot].store(lds, std::memory_order_relaxed);
        }
    }
    int pre_moved = 0;
    if (continuous && action != nullptr) {
        hipLaunchKernelGGL(maze_cont_move_kernel, dim3((n + MZ_BLOCK - 1) / MZ_BLOCK), dim3(MZ_BLOCK), 0,
                           (hipStream_t)stream, *T, *st, vk.col_dist, n, static_cast<const float *>(action));
        if (int rc = mg::check_launch("maze_cont_move_kernel")) return rc;
        pre_moved = 1;
    }
    // The stock configuration (maze3d_step_kernel<., true>): the caller vouches for ONE cell size for the whole task table
    // (mg_maze_view.uniform_cell_size) and every flag the general kernel would evaluate per env comes out "power of two, integer
    // texel addressing" — evaluated here exactly as the kernel evaluates them.
    bool stock = false;
    // the ABI is re-entrant — the environment is read once, not on every step
    static const bool force_generic = getenv("MG_MAZE3D_GENERIC") != nullptr;
    if (view->uniform_cell_size > 0.0 && !force_generic) {
        const double cs = view->uniform_cell_size, ttc = vk.text_size / cs, inv_ttc = 1.0 / ttc;
        int e2;
        stock = frexp(cs, &e2) == 0.5 && frexp(ttc, &e2) == 0.5 && vk.text_size_pow2 && (vk.TS & (vk.TS - 1)) == 0 && inv_ttc >= 1.0 &&
                (vk.eff_max + cs * (double)T->n) * ((double)vk.TS * vk.inv_text_size) < 1073741824.0;
    }
    // (MG_MAZE3D_NO_SMALL=1 keeps the general instantiation for one-wave frames too, for A/B timing; the frames are the same bit
    // for bit either way — same arithmetic, another route for the column record.)
#define MG_MAZE3D_LAUNCH(REC_, STOCK_, SMALL_, U8_)                                                                           \
    hipLaunchKernelGGL((maze3d_step_kernel<REC_, STOCK_, SMALL_, U8_>), dim3(n), dim3(n_waves * mg::WAVE), lds, (hipStream_t)stream, *T, *st, vk, \
                       task_type, max_steps, continuous, pre_moved, auto_reset, n, action, obs, reward, reward64, done)
#define MG_MAZE3D_PICK(REC_)                                                                                                  \
    do {                                                                                                                      \
        if (small) {                                                                                                          \
            if (stock && vk.obs_u8) MG_MAZE3D_LAUNCH(REC_, true, true, true);                                                 \
            else if (stock) MG_MAZE3D_LAUNCH(REC_, true, true, false);                                                        \
            else MG_MAZE3D_LAUNCH(REC_, false, true, false);                                                                  \
        } else {                                                                                                              \
            if (stock && vk.obs_u8) MG_MAZE3D_LAUNCH(REC_, true, false, true);                                                \
            else if (stock) MG_MAZE3D_LAUNCH(REC_, true, false, false);                                                       \
            else MG_MAZE3D_LAUNCH(REC_, false, false, false);                                                                 \
        }                                                                                                                     \
    } while (0)
    if (rec == 1) MG_MAZE3D_PICK(1); else MG_MAZE3D_PICK(2);
#undef MG_MAZE3D_PICK
#undef MG_MAZE3D_LAUNCH
    return mg::check_launch("maze3d_step_kernel");
}
