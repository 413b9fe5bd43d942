/*
 * oracle/maze_oracle.c — CPU restatement of the reference MetaMaze hot path.
 *
 * TEST INFRASTRUCTURE. The checker, never the product (see oracle/__init__.py).
 *
 * What it restates (paths relative to /root/reference/metagym/metamaze/envs):
 *   mo_reset            <- MazeBase.reset                      maze_base.py:40-63
 *   mo_evaluation_rule  <- MazeBase.evaluation_rule            maze_base.py:65-95, :191-192
 *   mo_step_2d          <- MazeCore2D.do_action                maze_2d.py:21-34 (+ DISCRETE_ACTIONS maze_env.py:14)
 *   mo_observe_2d       <- MazeCore2D.update_observation       maze_2d.py:89-121
 *   mo_step_disc3d      <- MazeCoreDiscrete3D.turn/move/do_action  maze_discrete_3d.py:51-81
 *   mo_step_cont3d      <- MazeCoreContinuous3D.do_action      maze_continuous_3d.py:47-56
 *                          + dynamics.py: nearest_point:17-29, collision_force:32-56,
 *                            vector_move:59-69, vector_move_with_collision:71-92
 *   mo_maze_view        <- maze_view                           ray_caster_utils.py:66-209
 *   dda_2d              <- DDA_2D                              ray_caster_utils.py:11-62
 *   mo_observe_3d       <- update_observation (+ life bar)     maze_discrete_3d.py:113-127
 *
 * Pinning: tests/test_oracle_maze.py checks every function against tests/golden/maze*.npz, recorded
 * from the unmodified reference by oracle/gen_golden_maze.py.
 *
 * Typing. The reference's renderer and dynamics are @njit functions: inside them numba types python
 * floats and literals as float64, float32 array elements as float32, and mixed ops as float64. The
 * goldens were produced with a numba stand-in that reproduces those rules (oracle/refstubs/numba).
 * The f32 column tables (cos_hp/cos_abs/sin_abs, ray_caster_utils.py:78-80) and the f32 texture
 * array are the only float32 values in the renderer; everything else is float64 and every store
 * into the int32 frame buffer truncates toward zero. Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "maze_oracle.h"

#define IDX(t, i, j) ((i) * (t)->n + (j))

static const int DISCRETE_ACTIONS[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}}; /* maze_env.py:14 */

/* ---- episode logic ------------------------------------------------------------------------- */

void mo_reset(const mo_task *t, int task_type, mo_state *s) {
    s->grid[0] = t->start[0];
    s->grid[1] = t->start[1];
    /* get_cell_center maze_base.py:194-197 */
    s->loc[0] = (float)(t->start[0] * t->cell_size + 0.5 * t->cell_size);
    s->loc[1] = (float)(t->start[1] * t->cell_size + 0.5 * t->cell_size);
    s->ori = 0.0;
    s->ori_idx = 0;
    s->steps = 0;
    if (task_type == MO_SURVIVAL) {
        int nn = t->n * t->n;
        for (int c = 0; c < nn; ++c) {
            s->wait_refresh[c] = 0;
            s->cur_food[c] = t->food_rewards[c];
            s->revival[c] = t->food_interval[c];
        }
        s->life = t->initial_life;
    }
}

int mo_evaluation_rule(const mo_task *t, int task_type, int max_steps, mo_state *s, double *reward) {
    int done;
    s->steps += 1;                                                     /* :66 */
    const int g = IDX(t, s->grid[0], s->grid[1]);
    if (task_type == MO_SURVIVAL) {
        double r;
        if (s->cur_food[g] > 1.0e-2) {                                 /* :71-75 */
            r = s->cur_food[g];
            s->wait_refresh[g] = 1;
            s->cur_food[g] = 0.0;
        } else {
            r = 0.0;
        }
        s->life += r + t->step_reward;                                 /* :78 */
        if (t->max_life < s->life) s->life = t->max_life;              /* :79 min(life, max_life) */
        done = (s->life < 0.0) || (s->steps > max_steps - 1);          /* :80, :191-192 */
        int nn = t->n * t->n;
        for (int c = 0; c < nn; ++c) s->revival[c] -= s->wait_refresh[c];   /* :83 */
        for (int c = 0; c < nn; ++c)                                   /* :84-88 */
            if (s->revival[c] < 0) {
                s->cur_food[c] = t->food_rewards[c];
                s->revival[c] = t->food_interval[c];
                s->wait_refresh[c] = 0;
            }
        *reward = r;
    } else {
        int goal = (t->goal[0] == s->grid[0]) && (t->goal[1] == s->grid[1]);
        *reward = t->step_reward + goal * t->goal_reward;              /* :92 */
        done = goal || (s->steps > max_steps - 1);
    }
    return done;
}

int mo_step_2d(const mo_task *t, int task_type, int max_steps, mo_state *s, int action, double *reward) {
    int ti = s->grid[0] + DISCRETE_ACTIONS[action][0];                 /* maze_2d.py:24-25 */
    int tj = s->grid[1] + DISCRETE_ACTIONS[action][1];
    /* python indexing: a negative index wraps (border walls make this unreachable) */
    int wi = ti < 0 ? ti + t->n : ti, wj = tj < 0 ? tj + t->n : tj;
    if (t->walls[IDX(t, wi, wj)] < 1) {                                /* :27-29 */
        s->grid[0] = ti;
        s->grid[1] = tj;
    }
    return mo_evaluation_rule(t, task_type, max_steps, s, reward);
}

void mo_observe_2d(const mo_task *t, int task_type, const mo_state *s, int vg, float *obs) {
    /* maze_2d.py:89-121: (2v+1)^2 float32 window, -1 outside the map */
    const int w = 2 * vg + 1, n = t->n;
    for (int a = 0; a < w; ++a)
        for (int b = 0; b < w; ++b) {
            int x = s->grid[0] - vg + a, y = s->grid[1] - vg + b;
            float v = -1.0f;
            if (x >= 0 && x < n && y >= 0 && y < n) {
                v = (float)(-t->walls[IDX(t, x, y)]);                  /* :113 int -> f32 */
                if (task_type == MO_SURVIVAL)
                    v = (float)((double)v + s->cur_food[IDX(t, x, y)]);            /* :117 f32 += f64 */
                else
                    v = (float)((double)v + ((x == t->goal[0] && y == t->goal[1]) ? 1.0 : 0.0)); /* :120 */
            }
            obs[a * w + b] = v;
        }
    if (task_type == MO_SURVIVAL) obs[vg * w + vg] = (float)s->life;  /* :118 */
}

int mo_step_disc3d(const mo_task *t, int task_type, int max_steps, mo_state *s, int action, double *reward) {
    const int turn = DISCRETE_ACTIONS[action][0], step = DISCRETE_ACTIONS[action][1];
    s->ori_idx = ((s->ori_idx + turn) % 4 + 4) % 4;                    /* maze_discrete_3d.py:69-72 */
    int g0 = s->grid[0], g1 = s->grid[1];                              /* :51-67 */
    if (s->ori_idx == 0) g0 += step;
    else if (s->ori_idx == 1) g1 += step;
    else if (s->ori_idx == 2) g0 -= step;
    else g1 -= step;
    if (g0 >= 0 && g0 < t->n && g1 >= 0 && g1 < t->n && t->walls[IDX(t, g0, g1)] == 0) {
        s->grid[0] = g0;
        s->grid[1] = g1;
    }
    return mo_evaluation_rule(t, task_type, max_steps, s, reward);
}

/* ---- continuous dynamics, dynamics.py -------------------------------------------------------- */

static const float OFFSET_10[2] = {0.5f, 0.5f}, OFFSET_01[2] = {-0.5f, 0.5f};
static const float OFFSET_m0[2] = {-0.5f, -0.5f}, OFFSET_0m[2] = {0.5f, -0.5f};

static float sum2(float a, float b) { return a + b; }

static float nearest_point(const float *pos, const float *l1, const float *l2, float *np_out) {
    float u[2] = {l2[0] - l1[0], l2[1] - l1[1]};                       /* :18 */
    float edge_norm = sqrtf(sum2(u[0] * u[0], u[1] * u[1]));           /* :19 */
    double m = (1.0e-6 > (double)edge_norm) ? 1.0e-6 : (double)edge_norm;  /* max(1e-6, edge_norm) -> f64 */
    u[0] = (float)((double)u[0] / m);                                  /* :20 f32 array /= f64 */
    u[1] = (float)((double)u[1] / m);
    float dist_1 = sum2((pos[0] - l1[0]) * u[0], (pos[1] - l1[1]) * u[1]);   /* :22 */
    const float *p;
    float lp[2];
    if (dist_1 > edge_norm) p = l2;                                    /* :23-24 */
    else if (dist_1 < 0) p = l1;                                       /* :25-26 */
    else {                                                             /* :27-29 */
        lp[0] = l1[0] + dist_1 * u[0];
        lp[1] = l1[1] + dist_1 * u[1];
        p = lp;
    }
    float d0 = pos[0] - p[0], d1 = pos[1] - p[1];
    np_out[0] = p[0];
    np_out[1] = p[1];
    return sqrtf(sum2(d0 * d0, d1 * d1));
}

static void collision_force(const float *dv, double cell_size, double col_dist, float *out) {
    double dist = (double)sqrtf(sum2(dv[0] * dv[0], dv[1] * dv[1]));   /* :33 */
    double eff = col_dist / cell_size;                                 /* :34 */
    out[0] = out[1] = 0.0f;
    if (dist > 0.708 + eff) return;                                    /* :35-36 */
    if (fabsf(dv[0]) < 0.5f && fabsf(dv[1]) < 0.5f) {                  /* :37-38 */
        double mx = dist > 1.0e-6 ? dist : 1.0e-6;
        float k = (float)(0.50 / mx * (0.708 + eff - dist) * cell_size);
        out[0] = k * dv[0];
        out[1] = k * dv[1];
        return;
    }
    int x_pos = (dv[0] + dv[1]) > 0, y_pos = (dv[1] - dv[0]) > 0;      /* :39-40 */
    float np_[2], d;
    if (x_pos && y_pos) d = nearest_point(dv, OFFSET_10, OFFSET_01, np_);
    else if (!x_pos && y_pos) d = nearest_point(dv, OFFSET_01, OFFSET_m0, np_);
    else if (!x_pos && !y_pos) d = nearest_point(dv, OFFSET_m0, OFFSET_0m, np_);
    else d = nearest_point(dv, OFFSET_0m, OFFSET_10, np_);
    if (eff < (double)d) return;                                       /* :50-51 */
    float ori[2] = {dv[0] - np_[0], dv[1] - np_[1]};                   /* :53 */
    float on = sqrtf(sum2(ori[0] * ori[0], ori[1] * ori[1]));          /* :54 */
    double r = 1.0 / ((1.0e-6 > (double)on) ? 1.0e-6 : (double)on);    /* :55 */
    ori[0] = (float)((double)ori[0] * r);
    ori[1] = (float)((double)ori[1] * r);
    float k = (float)(0.50 * (eff - (double)d) * cell_size);           /* :56 */
    out[0] = k * ori[0];
    out[1] = k * ori[1];
}

int mo_step_cont3d(const mo_task *t, int task_type, int max_steps, double collision_dist, mo_state *s,
                   double turn, double walk, double *reward) {
    const double PI = 3.1415926, t_PI = 6.2831852;                     /* dynamics.py:6-7 */
    /* maze_continuous_3d.py:48-49 numpy.clip(x, -1, 1) */
    double turn_rate = (turn < -1 ? -1 : (turn > 1 ? 1 : turn)) * PI;
    double walk_speed = walk < -1 ? -1 : (walk > 1 ? 1 : walk);
    if (walk_speed < 0) walk_speed *= 0.50;                            /* dynamics.py:73-74 */
    double ori = s->ori;
    float pos[2] = {s->loc[0], s->loc[1]};
    const float cs32 = (float)t->cell_size;   /* f32 array / python float: the float is weak */
    for (int it = 0; it < (int)(100 * 0.10); ++it) {                   /* :76 */
        /* vector_move dynamics.py:59-69, all float64 */
        double fin = ori + turn_rate * 0.01;
        double off_ori = 0.5 * (fin + ori);
        double off = walk_speed * 0.01;
        float d_x = (float)(cos(off_ori) * off), d_y = (float)(sin(off_ori) * off);
        while (fin > t_PI) fin -= t_PI;
        while (fin < 0) fin += t_PI;
        ori = fin;
        float exp_pos[2] = {pos[0] + d_x, pos[1] + d_y};               /* :78 */
        float exp_cell[2] = {exp_pos[0] / cs32, exp_pos[1] / cs32};    /* :79 */
        float col[2] = {0.0f, 0.0f};
        for (int i = -1; i < 2; ++i)
            for (int j = -1; j < 2; ++j) {
                int w_i = i + (int)exp_cell[0], w_j = j + (int)exp_cell[1];   /* :85-86 */
                if (w_i > -1 && w_i < t->n && w_j > -1 && w_j < t->n && t->walls[IDX(t, w_i, w_j)] > 0) {
                    float cd[2];                                       /* :89 */
                    cd[0] = (exp_cell[0] - floorf(exp_cell[0])) - ((float)i + 0.5f);
                    cd[1] = (exp_cell[1] - floorf(exp_cell[1])) - ((float)j + 0.5f);
                    float f[2];
                    collision_force(cd, t->cell_size, collision_dist, f);
                    col[0] += f[0];
                    col[1] += f[1];
                }
            }
        pos[0] = col[0] + exp_pos[0];                                  /* :91 */
        pos[1] = col[1] + exp_pos[1];
    }
    s->ori = ori;
    s->loc[0] = pos[0];
    s->loc[1] = pos[1];
    /* get_loc_grid maze_base.py:199-202: np.float32 / python float -> f32, then int() */
    s->grid[0] = (int)(pos[0] / cs32);
    s->grid[1] = (int)(pos[1] / cs32);
    return mo_evaluation_rule(t, task_type, max_steps, s, reward);
}

/* ---- renderer --------------------------------------------------------------------------------- */

#define MAX_TRANSP 256

typedef struct { double dist; int i, j, side; double strength; } transp_hit;

static int to_int_clamped(double x, int lo, int hi) {
    /* int(x) followed by max(lo,.) / min(hi,.) without C's out-of-range UB */
    if (!(x > (double)lo - 1.0)) return lo;
    if (x >= (double)hi + 1.0) return hi + 1;
    return (int)x;
}

/* DDA_2D ray_caster_utils.py:11-62. cos_ori / sin_ori are the float32 table values. */
static double dda_2d(const mo_task *t, const double *pos, int i, int j, float cos_f, float sin_f,
                     const double *transp, double max_vision, int *hit_i, int *hit_j, int *hit_side,
                     transp_hit *list, int *n_list) {
    const int n = t->n;
    const double cs = t->cell_size, c = (double)cos_f, s = (double)sin_f;
    const double delta_x = fabs(c) < 1.0e-6 ? 1.0e+6 : fabs(cs / c);   /* :12-13 */
    const double delta_y = fabs(s) < 1.0e-6 ? 1.0e+6 : fabs(cs / s);
    const double d_x = c > 0 ? ((i + 1) * cs - pos[0]) : (i * cs - pos[0]);   /* :14-15 */
    const double d_y = s > 0 ? ((j + 1) * cs - pos[1]) : (j * cs - pos[1]);
    double side_x = fabs(c) < 1.0e-6 ? 1.0e+6 : d_x / c;               /* :16-17 */
    double side_y = fabs(s) < 1.0e-6 ? 1.0e+6 : d_y / s;
    const int di = c > 0 ? 1 : -1, dj = s > 0 ? 1 : -1;
    int hi = i, hj = j, side = 0;
    double hit_dist = 0.0;
    *n_list = 0;
    if (hi >= 0 && hi < n && hj >= 0 && hj < n && transp[IDX(t, hi, hj)] > 0.01) {   /* :25-29 */
        transp_hit h = {side_x < side_y ? side_x : side_y, hi, hj, side_x < side_y ? 0 : 1, transp[IDX(t, hi, hj)]};
        list[(*n_list)++] = h;
    }
    while (hit_dist < max_vision) {                                    /* :31-61 */
        if (side_x < side_y) {
            hi += di;
            side_y -= side_x;
            hit_dist += side_x;
            if (hi < 0 || hi >= n) {
                if (hj < 0 || hj >= n) { hit_dist = 1.0e+6; break; }
            } else if (hj >= 0 && hj < n) {
                if (transp[IDX(t, hi, hj)] > 0.01 && *n_list < MAX_TRANSP) {
                    transp_hit h = {hit_dist, hi, hj, 0, transp[IDX(t, hi, hj)]};
                    list[(*n_list)++] = h;
                }
                if (t->walls[IDX(t, hi, hj)] > 0) { side = 0; break; }
            }
            side_x = delta_x;
        } else {
            hj += dj;
            side_x -= side_y;
            hit_dist += side_y;
            if (hi < 0 || hi >= n) {
                if (hj < 0 || hj >= n) { hit_dist = 1.0e+6; break; }
            } else if (hj >= 0 && hj < n) {
                if (transp[IDX(t, hi, hj)] > 0.01 && *n_list < MAX_TRANSP) {
                    transp_hit h = {hit_dist, hi, hj, 1, transp[IDX(t, hi, hj)]};
                    list[(*n_list)++] = h;
                }
                if (t->walls[IDX(t, hi, hj)] > 0) { side = 1; break; }
            }
            side_y = delta_y;
        }
    }
    *hit_i = hi;
    *hit_j = hj;
    *hit_side = side;
    return hit_dist;
}

static void blend_transparent(int32_t *px, double tf) {
    /* (1 - tf) * rgb(int32) + tf * TRANSPARENT_RGB(0,255,0 f32) -> int32, ray_caster_utils.py:125 */
    const double T[3] = {0.0, 255.0, 0.0};
    for (int c = 0; c < 3; ++c) px[c] = (int32_t)((1.0 - tf) * (double)px[c] + tf * T[c]);
}

void mo_maze_view(const mo_task *t, const mo_view *v, const double pos[2], double s_ori, double c_ori,
                  const double *transp, int32_t *rgb) {
    const int H = v->H, V = v->V, n = t->n, TS = v->tex_size;
    const double cs = t->cell_size, l_focal = v->l_focal, max_vision = v->max_vision;
    const double vh = t->agent_height, ceil_h = t->wall_height, text_size = v->text_size;
    const double half_h = v->tan_half_fov * l_focal;                    /* :68 */
    const double half_v = half_h * V / H;                               /* :69 */
    const double pixel_size = 2.0 * half_h / H;                         /* :70 */
    const double text_to_cell = text_size / cs;                         /* :73 */
    const double pixel_factor = pixel_size / l_focal;                   /* :76 */

    memset(rgb, 0, sizeof(int32_t) * (size_t)H * V * 3);
    float *transparent_array = (float *)calloc((size_t)H * V, sizeof(float));
    float *cos_hp = (float *)malloc(sizeof(float) * H), *cos_abs = (float *)malloc(sizeof(float) * H);
    float *sin_abs = (float *)malloc(sizeof(float) * H);
    double tan_hp = (-0.5 - H / 2.0) * pixel_factor;                    /* :83 (H / 2 is true division) */
    for (int d_h = 0; d_h < H; ++d_h) {                                 /* :84-90 */
        tan_hp += pixel_factor;
        double chp = sqrt(1.0 / (1.0 + tan_hp * tan_hp));
        double shp = tan_hp * chp;
        sin_abs[d_h] = (float)(shp * c_ori + chp * s_ori);
        cos_abs[d_h] = (float)(chp * c_ori - shp * s_ori);
        cos_hp[d_h] = (float)chp;
    }

    /* floor :94-126 */
    for (int d_v = V - 1; d_v > V / 2; --d_v) {
        double v_screen = (d_v + 0.5) * pixel_size - half_v;
        double distance = vh / v_screen * l_focal;
        double light = v_screen / l_focal;
        if (distance > max_vision) continue;
        for (int d_h = 0; d_h < H; ++d_h) {
            double eff = distance / (double)cos_hp[d_h];
            double a = 2.0 * eff / max_vision - 1.0;
            a = a > 0.0 ? a : 0.0;
            a = a < 1.0 ? a : 1.0;
            double alpha = a * light;
            double hit_x = eff * (double)cos_abs[d_h] + pos[0];
            double hit_y = eff * (double)sin_abs[d_h] + pos[1];
            double fi = hit_x / cs, fj = hit_y / cs;
            double d_i = fi - floor(fi), d_j = fj - floor(fj);
            int i = to_int_clamped(fi, -2, n + 1), j = to_int_clamped(fj, -2, n + 1);
            if (i < n && i >= 0 && j < n && j >= 0) {
                int text_id = t->texts[IDX(t, i, j)];
                d_i /= text_to_cell;
                d_j /= text_to_cell;
                d_i -= floor(d_i);
                d_j -= floor(d_j);
                d_i *= TS;
                d_j *= TS;
                const float *tex = &v->textures[(((size_t)text_id * TS + (int)d_i) * TS + (int)d_j) * 3];
                int32_t *px = &rgb[((size_t)d_h * V + d_v) * 3];
                for (int c = 0; c < 3; ++c)
                    px[c] = (int32_t)(light * (alpha * 0.0 + (1.0 - alpha) * (double)tex[c]));
                if (transp[IDX(t, i, j)] > 0.01) {
                    blend_transparent(px, transp[IDX(t, i, j)] * 0.50 + 0.10);
                    transparent_array[(size_t)d_h * V + d_v] = 1.0f;
                }
            }
        }
    }

    /* ceiling :128-153 */
    for (int d_v = 0; d_v < V / 2; ++d_v) {
        double v_screen = half_v - (d_v + 0.5) * pixel_size;
        double distance = (ceil_h - vh) / v_screen * l_focal;
        double light = v_screen / l_focal;
        if (distance > max_vision) continue;
        for (int d_h = 0; d_h < H; ++d_h) {
            double eff = distance / (double)cos_hp[d_h];
            double alpha = 2.0 * eff / max_vision - 1.0;
            alpha = alpha > 0.0 ? alpha : 0.0;
            alpha = alpha < 1.0 ? alpha : 1.0;
            double hit_x = eff * (double)cos_abs[d_h] + pos[0];
            double hit_y = eff * (double)sin_abs[d_h] + pos[1];
            int t_i = to_int_clamped(hit_x / cs, -2, n + 1), t_j = to_int_clamped(hit_y / cs, -2, n + 1);
            double fi = hit_x / text_size, fj = hit_y / text_size;
            double d_i = fi - floor(fi), d_j = fj - floor(fj);
            d_i *= TS;
            d_j *= TS;
            const uint8_t *tex = &v->ceil_tex[(((size_t)(int)d_i) * TS + (int)d_j) * 3];
            int32_t *px = &rgb[((size_t)d_h * V + d_v) * 3];
            for (int c = 0; c < 3; ++c)
                px[c] = (int32_t)(light * (alpha * 0.0 + (1.0 - alpha) * (double)tex[c]));
            if (t_i >= 0 && t_i < n && t_j >= 0 && t_j < n && transp[IDX(t, t_i, t_j)] > 0) {
                blend_transparent(px, transp[IDX(t, t_i, t_j)] * 0.50 + 0.10);
                transparent_array[(size_t)d_h * V + d_v] = 1.0f;
            }
        }
    }

    /* walls :155-205 */
    transp_hit list[MAX_TRANSP];
    for (int d_h = 0; d_h < H; ++d_h) {
        int i = (int)(pos[0] / cs), j = (int)(pos[1] / cs);
        int hit_i, hit_j, hit_side, n_list;
        double hit_dist = dda_2d(t, pos, i, j, cos_abs[d_h], sin_abs[d_h], transp, max_vision, &hit_i, &hit_j,
                                 &hit_side, list, &n_list);
        if (hit_dist > max_vision) continue;
        double alpha = 2.0 * hit_dist / max_vision - 1.0;
        alpha = alpha > 0.0 ? alpha : 0.0;
        alpha = alpha < 1.0 ? alpha : 1.0;
        int ci = hit_i < 0 ? hit_i + n : hit_i, cj = hit_j < 0 ? hit_j + n : hit_j;   /* python wrap */
        if (ci < 0 || ci >= n || cj < 0 || cj >= n) continue;   /* IndexError in the reference */
        int text_id = t->texts[IDX(t, ci, cj)];
        double hpx = hit_dist * (double)cos_abs[d_h] + pos[0];
        double hpy = hit_dist * (double)sin_abs[d_h] + pos[1];
        double local_h;
        float light;
        if (hit_side == 0) {
            local_h = hpy / cs;
            local_h -= floor(local_h);
            light = fabsf(cos_abs[d_h]);
        } else {
            local_h = hpx / cs;
            local_h -= floor(local_h);
            light = fabsf(sin_abs[d_h]);
        }
        double ratio = hit_dist * (double)cos_hp[d_h] / l_focal;
        double top_v = (ceil_h - vh) / ratio, bot_v = vh / ratio;
        int v_s = to_int_clamped((half_v - top_v) / pixel_size, 0, V);
        int v_e = to_int_clamped((half_v + bot_v) / pixel_size, -1, V - 1);
        if (v_s < 0) v_s = 0;
        if (v_e > V) v_e = V;
        for (int d_v = v_s; d_v < v_e; ++d_v) {
            double local_v = (half_v - (d_v + 0.5) * pixel_size) * ratio + vh;
            double d_i = local_h / text_size, d_j = local_v / text_size;
            d_i -= floor(d_i);
            d_j -= floor(d_j);
            int ti = (int)(TS * d_i), tj = (int)(TS * d_j);
            const float *tex = &v->textures[(((size_t)text_id * TS + ti) * TS + tj) * 3];
            int32_t *px = &rgb[((size_t)d_h * V + d_v) * 3];
            for (int c = 0; c < 3; ++c)
                px[c] = (int32_t)((double)light * (alpha * 0.0 + (1.0 - alpha) * (double)tex[c]));
        }
        for (int k = 0; k < n_list; ++k) {                              /* :194-205 */
            double r2 = list[k].dist * (double)cos_hp[d_h] / l_focal;
            double tf = list[k].strength * 0.50 + 0.10;
            double tv = (ceil_h - vh) / r2, bv = vh / r2;
            int s2 = to_int_clamped((half_v - tv) / pixel_size, 0, V);
            int e2 = to_int_clamped((half_v + bv) / pixel_size, -1, V - 1);
            if (s2 < 0) s2 = 0;
            if (e2 > V) e2 = V;
            for (int d_v = s2; d_v < e2; ++d_v)
                if (transparent_array[(size_t)d_h * V + d_v] < 1)
                    blend_transparent(&rgb[((size_t)d_h * V + d_v) * 3], tf);
        }
    }
    free(transparent_array);
    free(cos_hp);
    free(cos_abs);
    free(sin_abs);
}

static void py_slice_bounds(long a, long b, long len, long *lo, long *hi) {
    if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;
    if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
    *lo = a; *hi = b;
}

void mo_observe_3d(const mo_task *t, int task_type, const mo_view *v, const mo_state *s, int continuous,
                   const float *ori_sin4, const float *ori_cos4, int32_t *rgb) {
    const int nn = t->n * t->n;
    double *transp = (double *)malloc(sizeof(double) * nn);
    if (task_type == MO_SURVIVAL) memcpy(transp, s->cur_food, sizeof(double) * nn);   /* alias, maze_base.py:57 */
    else {
        for (int c = 0; c < nn; ++c) transp[c] = 0.0;
        transp[IDX(t, t->goal[0], t->goal[1])] = 1.0;                   /* maze_base.py:59-60 */
    }
    double pos[2], so, co;
    if (continuous) {
        pos[0] = (double)s->loc[0];
        pos[1] = (double)s->loc[1];
        so = sin(s->ori);
        co = cos(s->ori);
    } else {
        pos[0] = s->grid[0] * t->cell_size + 0.5 * t->cell_size;        /* get_cell_center */
        pos[1] = s->grid[1] * t->cell_size + 0.5 * t->cell_size;
        so = (double)ori_sin4[s->ori_idx];
        co = (double)ori_cos4[s->ori_idx];
    }
    mo_maze_view(t, v, pos, so, co, transp, rgb);
    if (task_type == MO_SURVIVAL) {                                     /* maze_discrete_3d.py:118-126 */
        double lifebar_l = s->life / t->max_life * (0.80 * v->V);
        double sx = 0.10 * v->V, sy = 0.10 * v->V;
        long start_x = (long)sx, start_y = (long)sy;
        long end_x = (long)(sx + lifebar_l), end_y = (long)(sy + 0.05 * v->H);
        long x0, x1, y0, y1;
        py_slice_bounds(start_x, end_x, v->H, &x0, &x1);
        py_slice_bounds(start_y, end_y, v->V, &y0, &y1);
        for (long x = x0; x < x1; ++x)
            for (long y = y0; y < y1; ++y) {
                int32_t *px = &rgb[((size_t)x * v->V + y) * 3];
                px[0] = 255; px[1] = 0; px[2] = 0;
            }
    }
    free(transp);
}
