/* oracle/walker_oracle.h — C restatement of the articulated-body env step. TEST INFRASTRUCTURE (see walker_oracle.c). */
#ifndef WALKER_ORACLE_H
#define WALKER_ORACLE_H
#include <stdint.h>

#define WO_MAX_BODIES 16
#define WO_MAX_JOINTS 24
#define WO_MAX_DOF (6 + WO_MAX_JOINTS)
#define WO_MAX_SPHERES 128
#define WO_MAX_FEET 6
#define WO_MAX_GEOMS 24
#define WO_MAX_PAIRS 128
#define WO_MAX_CONTACTS 12
#define WO_MAX_CANDIDATES 48       /* abd.MAX_CANDIDATES, walker.hip W_MAXCAND */
#define WO_MAX_ROWS (3 * WO_MAX_CONTACTS + WO_MAX_JOINTS)

typedef struct wo_model {            /* topology (mg_walker_topology) + one row of the model table (mg_walker_models) */
    int32_t nb, nj, ns, nf, ng, npairs;
    int32_t body_parent[WO_MAX_BODIES], joint_body[WO_MAX_JOINTS], sphere_body[WO_MAX_SPHERES], foot_body[WO_MAX_FEET],
        geom_body[WO_MAX_GEOMS];
    uint8_t pair_a[WO_MAX_PAIRS], pair_b[WO_MAX_PAIRS];
    const double *table;
    const double *sph_margin;        /* [ns] per-proxy contact margin (mjcf.contact_margins) or NULL: wo_params.contact_margin for all */
} wo_model;

typedef struct wo_params {
    double dt;                       /* 0.005 */
    int32_t substeps, iterations;    /* 4, 5 */
    double erp, limit_erp, gravity, friction, self_friction;
    int32_t self_collision, max_steps;
    double alive_z, alive_bonus, initial_z, walk_target_x, walk_target_y;
    int32_t initial_z_from_state;    /* ant: initial_z = first calc_state's z (walker_base.py:44-45) */
    int32_t floor_in_parts, torque_f32, height_f32;
    double body_linear_damping, body_angular_damping;   /* btMultiBody velocity damping of every body (0.04 / 0.04 in the "bullet" preset) */
    double max_coordinate_velocity;                     /* btMultiBody's clamp of every generalized velocity (100 in the "bullet" preset; 0 = off) */
    double contact_margin;                              /* contact margin of every proxy when wo_model.sph_margin is NULL (0 = penetration only): abd.Params.contact_margin */
} wo_params;

typedef struct wo_state { double pos[3], rot[9], vel[3], omega[3], q[WO_MAX_JOINTS], qd[WO_MAX_JOINTS]; } wo_state;

typedef struct wo_env {
    wo_state s;
    double potential, initial_z;
    float feet_contact[WO_MAX_FEET];
    int32_t steps, floor_known, initial_z_unset;
} wo_env;

int wo_substep(const wo_model *m, const wo_params *prm, wo_state *s, const double *tau_motor, unsigned long long touch[2]);
void wo_env_reset(const wo_model *m, const wo_params *prm, wo_env *e, const double *joint_noise, float *obs);
int wo_env_step(const wo_model *m, const wo_params *prm, wo_env *e, const float *action, float *obs, double *reward, double *rewards5);
long wo_run(const wo_model *models, const int *task_id, const wo_params *prm, wo_env *envs, int n_envs, int n_steps,
            const float *actions, int n_action_rows);
int wo_flops_read(unsigned long long *out6, int clear);   /* 1 when built with -DWO_COUNT_FLOPS */
#endif
