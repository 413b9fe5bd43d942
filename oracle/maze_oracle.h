/* oracle/maze_oracle.h — TEST INFRASTRUCTURE (see maze_oracle.c). */
#ifndef MAZE_ORACLE_H
#define MAZE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MO_ESCAPE = 0, MO_SURVIVAL = 1 };

/* TaskConfig, maze_task.py:15-17 (arrays are [n][n], first index = x like the reference) */
typedef struct {
    int32_t n;
    int32_t start[2], goal[2];
    const int32_t *walls;        /* cell_walls */
    const int32_t *texts;        /* cell_texts (0 ground, 1..6 wall textures) */
    const double *food_rewards;
    const int32_t *food_interval;
    double cell_size, wall_height, agent_height;
    double initial_life, max_life, step_reward, goal_reward;
} mo_task;

/* MazeBase per-episode state, maze_base.py:40-63 (+ the 3-D cores' orientation / location) */
typedef struct {
    int32_t grid[2];
    int32_t steps;
    int32_t ori_idx;       /* discrete 3-D heading index, maze_discrete_3d.py:46-48 */
    double ori;            /* continuous 3-D heading (np.float64) */
    float loc[2];          /* continuous 3-D location (float32 array) */
    double life;
    double *cur_food;      /* [n*n] SURVIVAL only */
    int32_t *wait_refresh; /* [n*n] */
    int32_t *revival;      /* [n*n] */
} mo_state;

/* renderer constants, maze_discrete_3d.py:18-37,113-117 and the host-prepared column tables */
typedef struct {
    int32_t H, V;                /* resolution_horizon, resolution_vertical */
    double max_vision, l_focal, text_size, tan_half_fov;
    const float *textures;       /* [T][64][64][3] float32 (integral values) */
    const uint8_t *ceil_tex;     /* [64][64][3] */
    int32_t tex_size;            /* 64 */
} mo_view;

void mo_reset(const mo_task *t, int task_type, mo_state *s);
int mo_evaluation_rule(const mo_task *t, int task_type, int max_steps, mo_state *s, double *reward);
int mo_step_2d(const mo_task *t, int task_type, int max_steps, mo_state *s, int action, double *reward);
void mo_observe_2d(const mo_task *t, int task_type, const mo_state *s, int view_grid, float *obs);
int mo_step_disc3d(const mo_task *t, int task_type, int max_steps, mo_state *s, int action, double *reward);
int mo_step_cont3d(const mo_task *t, int task_type, int max_steps, double collision_dist, mo_state *s,
                   double turn, double walk, double *reward);
/* transparents: f64 [n*n] (SURVIVAL: the live food array; ESCAPE: one-hot goal as 0/1) */
void mo_maze_view(const mo_task *t, const mo_view *v, const double pos[2], double s_ori, double c_ori,
                  const double *transparents, int32_t *rgb /* [H][V][3] */);
void mo_observe_3d(const mo_task *t, int task_type, const mo_view *v, const mo_state *s, int continuous,
                   const float *ori_sin4, const float *ori_cos4, int32_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
