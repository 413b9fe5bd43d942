/* oracle/quadrotor_oracle.h — TEST INFRASTRUCTURE (see quadrotor_oracle.c). */
#ifndef QUADROTOR_ORACLE_H
#define QUADROTOR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { QO_TASK_NO_COLLISION = 0, QO_TASK_VELOCITY = 1, QO_TASK_HOVERING = 2 };

typedef struct {
    /* python floats (doubles) exactly as _parse_cfg keeps them, quadrotorsim.py:50-109 */
    double precision, quality;
    double ct0, ct1, ct2, mm, jm, ra, phi;
    double fail_velocity, fail_w, fail_range;
    double min_voltage, max_voltage;
    /* env-level scalars, env.py:46-114 */
    double dt, healthy_reward, z_offset;
    int64_t x_offset, y_offset;
    int32_t nt, task;
    /* float32 arrays */
    float inertia_inv[9], drag_m[9], drag_f[9], gravity_center[3], prop_coord[12];
    /* height map (row-major [map_h][map_w]); NULL = the default all-zero 100x100 map */
    const int32_t *map;
    int32_t map_h, map_w;
    /* velocity_control: target trajectory float32 [nt][3] (define_velocity_control_task) */
    const float *velocity_targets;
} qo_consts;

typedef struct {
    float pos[3];
    double vel[3];
    double omega[3];
    float propw[4];
    float R[9];
    float Rinv[9];   /* _coordination_converter_to_body; always inv(R) */
    float power;
    float pos0_z;    /* env.pos_0[2], set at reset (env.py:123) */
} qo_state;

void qo_default_consts(qo_consts *c);
void qo_zero_state(qo_state *s);
void qo_refresh_inverse(qo_state *s);
void qo_inv3_f32(const float *A, float *Ainv);
void qo_substep(const qo_consts *c, qo_state *s, const double act[4]);
int qo_failed(const qo_consts *c, const qo_state *s);
int qo_sim_step(const qo_consts *c, qo_state *s, const float act[4]);
void qo_observe(const qo_consts *c, const qo_state *s, float obs[16]);
int qo_env_step(const qo_consts *c, qo_state *s, int *ct, const float act[4],
                float obs[16], double *reward, int *done);
void qo_batch_env_step(const qo_consts *c, int n, qo_state *states, int *ct, const float *actions,
                       float *obs, double *reward, int *done, int *failed);
void qo_velocity_targets(const qo_consts *c, int nt, const float *actions, float *targets);
int qo_env_step_velocity(const qo_consts *c, qo_state *s, int *ct, const float act[4], float obs[19],
                         double *reward, int *done);
long qo_batch_run(const qo_consts *c, int n, qo_state *states, const qo_state *init, int *ct,
                  const float *actions, int n_batches, int iters);
/* ---- fused auto-reset (not a reference operation: restates metagym_amd's device-side reset, whose
 * zero-state / noise formula follows QuadrotorSim.reset quadrotorsim.py:239-258) ------------------------ */
typedef struct {
    float init_velocity[3], init_angular_velocity[3];
    double init_velocity_noisy, init_angular_velocity_noisy;
    uint64_t seed, env_id_base;
} qo_autoreset;

void qo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void qo_reset_noise(const qo_autoreset *ar, uint64_t gid, uint32_t episode, double vel[3], double omega[3]);
void qo_reset_random(const qo_autoreset *ar, qo_state *s, uint64_t gid, uint32_t episode);
void qo_batch_env_step_autoreset(const qo_consts *c, const qo_autoreset *ar, int n, qo_state *states, int *ct,
                                 uint32_t *episode, const float *actions, float *obs, double *reward, int *done,
                                 int *failed);
size_t qo_sizeof_state(void);
size_t qo_sizeof_consts(void);

void qo_set_legacy_promotion(int on);   /* numpy 1.22 reading of the python-float x float32-scalar mixes; tests only */

#ifdef __cplusplus
}
#endif
#endif
