/*
 * metagym_hip.h — C ABI of libmetagym_hip.so, the MI355X (gfx950) batched environment engine.
 *
 * The reference (PaddlePaddle/MetaGym) has no native layer and no FFI: every env is a Python
 * object simulating ONE environment. This ABI is therefore new; each entry point names the
 * reference Python method whose body it replaces for N environments at once. The host-side
 * mirror of the reference's gym.Env classes lives in metagym_amd/ and calls these through ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes. No torch / HIP types in any signature; `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream).
 *   - Every `*_d` / state / io pointer is a DEVICE pointer into caller-owned memory (torch-ROCm
 *     tensors in practice). The library allocates nothing persistent: state_dict()/checkpointing
 *     is a tensor clone on the caller's side.
 *   - All state is structure-of-arrays: component c of env e is at base[c * n_envs + e], so
 *     lane e of a wavefront touches consecutive addresses (coalesced HBM access).
 *   - Calls are asynchronous: kernels are enqueued on `stream` and the call returns without
 *     synchronising. Re-entrant; no global mutable state except the thread-local error string.
 *   - Multi-GPU processes: every launching entry point looks up the HIP device that owns the state memory it
 *     is handed (hipPointerGetAttributes; cached inside a plan) and makes it current for the duration of the
 *     call when it is not already, so `stream` must belong to that device.
 *   - Return value: MG_OK (0) or a negative error code. hipError_t values are returned negated;
 *     argument errors are in the -1000 range. Nothing throws or aborts across the ABI.
 *   - Per-environment simulation failures (the reference `raise`s out of step()) are DATA, not
 *     errors: they come back in the `failed` byte array.
 */
#ifndef METAGYM_HIP_H
#define METAGYM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 7

#define MG_OK 0
#define MG_ERR_NULL_POINTER (-1001)
#define MG_ERR_BAD_SIZE (-1002)
#define MG_ERR_BAD_CONFIG (-1003)
#define MG_ERR_UNSUPPORTED (-1004)

/* ABI version of the loaded library (== MG_ABI_VERSION of the header it was built from). */
int mg_abi_version(void);
/* Human-readable description of the last error on the calling thread ("" if none). */
const char *mg_last_error(void);
/* Name of the device architecture the kernels were compiled for ("gfx950"). */
const char *mg_target_arch(void);

/* Self-test hook: out_d[i][0..3] = Philox4x32-10(counter = ctr_key_d[i][0..3], key = ctr_key_d[i][4..5]) computed by
 * the device function every fused auto-reset draws its noise from (DEVICE u32 [n][6] -> DEVICE u32 [n][4]).
 * tests/ checks it against the published Random123 known-answer vectors. */
int mg_selftest_philox(const uint32_t *ctr_key_d, uint32_t *out_d, int32_t n, void *stream);

/* ========================================================================================
 * Quadrotor — replaces metagym/quadrotor/quadrotorsim.py + env.py for N envs
 * ======================================================================================== */

enum { MG_QUADROTOR_TASK_NO_COLLISION = 0, MG_QUADROTOR_TASK_VELOCITY_CONTROL = 1,
       MG_QUADROTOR_TASK_HOVERING_CONTROL = 2 };

/* Physical + task constants: the parsed metagym/quadrotor/config.json (quadrotorsim.py:50-109,
 * "python floats" stay doubles, float32 matrices stay float32) and the Quadrotor.__init__
 * arguments (env.py:46-114). Uniform across envs; passed to the kernel by value (scalar regs). */
typedef struct mg_quadrotor_config {
    double precision;        /* cfg['precision']: Euler sub-step, s (0.001) */
    double quality;          /* cfg['quality']: mass, kg (0.5) */
    double ct0, ct1, ct2;    /* cfg['thrust']['CT'] */
    double mm, jm, ra, phi;  /* cfg['thrust'] Mm, Jm, RA, phi */
    double fail_velocity, fail_w, fail_range;   /* cfg['fail'] */
    double min_voltage, max_voltage;            /* cfg['electric'] */
    double dt;               /* env step, s (0.01); sub-steps per step = int(dt / precision) */
    double healthy_reward;   /* env.py:53 */
    double z_offset;         /* env.py:113 (5.0) */
    int64_t x_offset, y_offset;  /* start cell of the map, env.py:109-112 */
    int32_t nt;              /* episode length, env.py:48 */
    int32_t task;            /* MG_QUADROTOR_TASK_* */
    float inertia[9];        /* row-major cfg['inertia'] (inverted in float32 by the library) */
    float drag_m[9];         /* diag(cfg['drag'] m_xx, m_yy, m_zz) */
    float drag_f[9];         /* diag(cfg['drag'] f_xx, f_yy, f_zz) */
    float gravity_center[3];
    float prop_coord[12];    /* 4 propellers x (x, y, z) */
    const int32_t *map_d;    /* DEVICE int32[map_h][map_w] obstacle map, or NULL = flat floor */
    int32_t map_h, map_w;
    const float *velocity_targets_d;   /* DEVICE f32 [nt][3], task VELOCITY_CONTROL only: the trajectory
                                          of mg_quadrotor_velocity_targets (env.py:99-102) */
} mg_quadrotor_config;

/* Per-env simulator state (QuadrotorSim._zero_state quadrotorsim.py:20-28 + Quadrotor.ct env.py:66),
 * in the dtypes the reference holds after reset(): position f32, velocity f64, body rate f64,
 * propeller speed f32, rotation matrix f32. 116 bytes per env. The body<-world matrix
 * (`_coordination_converter_to_body`) is not state: it is inv(R) and is recomputed on load. */
typedef struct mg_quadrotor_state {
    float *pos;      /* [3][n] global_position */
    double *vel;     /* [3][n] global_velocity */
    double *omega;   /* [3][n] body_angular_velocity */
    float *propw;    /* [4][n] propeller_angular_velocity */
    float *rot;      /* [9][n] rotation_matrix, row-major index */
    int32_t *ct;     /* [n]    Quadrotor.ct step counter */
    uint32_t *episode;   /* [n] number of fused auto-resets env e has gone through = the Philox counter of its next
                            one (see mg_quadrotor_autoreset). Not a reference quantity; only read / written by the
                            auto-reset launches and may be NULL for every other call. */
} mg_quadrotor_state;

/* Fill `cfg` with the values of the reference's default config.json + default constructor args
 * (task hovering_control is NOT the reference default; set cfg->task yourself). */
int mg_quadrotor_default_config(mg_quadrotor_config *cfg);

/* Quadrotor.reset() env.py:116-125 + QuadrotorSim.reset() quadrotorsim.py:239-258 for every env
 * with mask[e] != 0 (mask == NULL: all). State is zeroed, then
 *   velocity  <- init_vel[c][e]   (f64 [3][n]; NULL: zeros)
 *   body rate <- init_omega[c][e] (f64 [3][n]; NULL: zeros)
 * The reference draws that noise from numpy's *global* RNG; the caller supplies it (the Python
 * layer reproduces the reference's draw order). ct is NOT cleared (env.py:116-125 does not).
 * obs (f32 [n][16], may be NULL) receives the reset observation for the selected envs. */
int mg_quadrotor_reset(const mg_quadrotor_config *cfg, int32_t n_envs, const mg_quadrotor_state *state,
                       const uint8_t *mask, const double *init_vel, const double *init_omega,
                       float *obs, void *stream);

/* Quadrotor.step(action) env.py:127-165 for all n envs: int(dt/precision) Euler sub-steps of
 * QuadrotorSim._run_internal (quadrotorsim.py:122-210) with the failure check after each one,
 * then sensors/state (quadrotorsim.py:260-293), collision (env.py:248-260), reward (env.py:211-246)
 * and the done rule (env.py:144-161).
 *   action   f32 [n][4]  motor voltages (clamped to [min_voltage, max_voltage] like the reference)
 *   obs      f32 [n][16] env.py:193-209 key order; [n][19] for VELOCITY_CONTROL (+ next_target_g_v_x/y/z)
 *   reward   f32 [n]     (may be NULL)
 *   reward64 f64 [n]     the reference returns a python float; optional exact copy (may be NULL)
 *   done     u8  [n]
 *   failed   u8  [n]     0 = ok; 1/2/3 = position / velocity / body-rate limit exceeded
 *                        (quadrotorsim.py:212-221). A failed env freezes at the failing sub-step,
 *                        reports done=1, reward=0 and must be reset. (may be NULL)
 * VELOCITY_CONTROL (env.py:150-157) has no map / collision test; its reward is
 * -min(dt*power, healthy) - 0.001 * |Rinv @ target[ct-1] - body velocity|_1 and needs
 * cfg->velocity_targets_d. */
int mg_quadrotor_step(const mg_quadrotor_config *cfg, int32_t n_envs, const mg_quadrotor_state *state,
                      const float *action, float *obs, float *reward, double *reward64,
                      uint8_t *done, uint8_t *failed, void *stream);

/* QuadrotorSim.define_velocity_control_task (quadrotorsim.py:306-319): nt env steps from the pre-reset
 * zero state, in which every simulator array is still float32 (so the whole sub-step runs in float32,
 * unlike after reset()), with the caller's action stream (DEVICE f32 [nt][4]; the reference draws
 * np.random.seed(seed); uniform(min_voltage, max_voltage, 4).astype(f32) per step). Writes the global
 * velocity after every step to targets_d (DEVICE f32 [nt][3]). One-off setup work, single lane. */
int mg_quadrotor_velocity_targets(const mg_quadrotor_config *cfg, int32_t nt, const float *actions_d,
                                  float *targets_d, void *stream);

/* n_steps consecutive env steps in ONE launch (state stays in registers between steps).
 *   action f32 [n_steps][n][4]; obs f32 [n_steps][n][16]; reward/done/failed [n_steps][n].
 * Semantically identical to calling mg_quadrotor_step n_steps times without resets in between
 * (the reference's own tests step on after done, tests/test_env.py). */
int mg_quadrotor_rollout(const mg_quadrotor_config *cfg, int32_t n_envs, int32_t n_steps,
                         const mg_quadrotor_state *state, const float *action, float *obs,
                         float *reward, double *reward64, uint8_t *done, uint8_t *failed, void *stream);

/* Fused episode management (NOT in the reference, where the user calls env.reset() after done):
 * like mg_quadrotor_rollout, but an env whose step ended with done=1 is reset inside the same
 * launch (QuadrotorSim.reset quadrotorsim.py:239-258: zero state + init noise) and the returned
 * observation row is the first observation of its next episode; reward/done/failed still describe
 * the step that ended. The noise of the k-th auto-reset of global env g is a pure function of
 * (seed, g, k): two Philox4x32-10 blocks with key = seed (lo, hi) and counter = (g lo, g hi, k, block),
 *   velocity[c]  = init_velocity[c]         + noisy_v * (w[c]   / 2^32) * (bit c   of w[6] ? +1 : -1)
 *   body rate[c] = init_angular_velocity[c] + noisy_w * (w[3+c] / 2^32) * (bit 3+c of w[6] ? +1 : -1)
 * with w[0..7] the eight output words, g = env_id_base + env index and k = state->episode[e], which the
 * launch increments. Nothing depends on a host-side step counter, so results are independent of how
 * envs are sharded across GPUs and of how many steps go into one launch, and the launch is
 * hipGraph-capturable as it stands (all arguments are replay-invariant). */
typedef struct mg_quadrotor_autoreset {
    float init_velocity[3];            /* cfg['init_velocity'] x, y, z */
    float init_angular_velocity[3];    /* cfg['init_angular_velocity'] x, y, z */
    double init_velocity_noisy;        /* cfg['init_velocity']['noisy'] (2.0) */
    double init_angular_velocity_noisy;/* cfg['init_angular_velocity']['noisy'] (5.0) */
    uint64_t seed;
    uint64_t env_id_base;              /* global id of env 0 of this shard (0 on a single GPU) */
} mg_quadrotor_autoreset;

int mg_quadrotor_step_autoreset(const mg_quadrotor_config *cfg, int32_t n_envs, int32_t n_steps,
                                const mg_quadrotor_state *state, const mg_quadrotor_autoreset *ar,
                                const float *action, float *obs, float *reward, double *reward64,
                                uint8_t *done, uint8_t *failed, void *stream);

/* Prepared stepping. Every entry point above folds `cfg` into kernel constants on each call (a few hundred
 * host instructions, two float32 sqrt searches). A plan does it once: mg_quadrotor_plan_init validates and
 * folds cfg (+ the optional auto-reset block), records the state pointers, n_envs and the HIP device that
 * owns the state memory into caller-owned HOST memory; mg_quadrotor_plan_step then only enqueues the launch
 * (and selects that device for the duration of the call if it is not the calling thread's current one).
 * Same kernels, same results as mg_quadrotor_step / _rollout / _step_autoreset with the same arguments.
 * The plan holds no resources and needs no destructor; it must be re-initialised when cfg, the state
 * tensors or n_envs change. */
typedef struct mg_quadrotor_plan { uint64_t opaque[128]; } mg_quadrotor_plan;

int mg_quadrotor_plan_init(mg_quadrotor_plan *plan, const mg_quadrotor_config *cfg,
                           const mg_quadrotor_autoreset *ar /* NULL = no fused reset */, int32_t n_envs,
                           const mg_quadrotor_state *state);
int mg_quadrotor_plan_step(const mg_quadrotor_plan *plan, int32_t n_steps, const float *action, float *obs,
                           float *reward, double *reward64, uint8_t *done, uint8_t *failed, void *stream);

/* ========================================================================================
 * MetaMaze — replaces metagym/metamaze/envs/{maze_base,maze_2d,maze_discrete_3d,
 *            maze_continuous_3d,dynamics,ray_caster_utils}.py for N envs
 * ======================================================================================== */

enum { MG_MAZE_ESCAPE = 0, MG_MAZE_SURVIVAL = 1 };

/* Task table: T TaskConfig tuples (maze_task.py:15-17) of identical size n x n, uploaded once by
 * the caller; env e plays task task_id[e]. Grids are [T][n][n] with the reference's index order
 * (first index = x). Read-only for the kernels. */
typedef struct mg_maze_tasks {
    int32_t n;                     /* cells per side */
    int32_t n_tasks;               /* T */
    const int32_t *start;          /* [T][2] */
    const int32_t *goal;           /* [T][2] */
    const int8_t *walls;           /* [T][n*n] cell_walls (0 free, 1 wall) */
    const uint8_t *texts;          /* [T][n*n] cell_texts (0 ground, 1.. wall textures) */
    const double *food_rewards;    /* [T][n*n] */
    const int32_t *food_interval;  /* [T][n*n] */
    const double *scalars;         /* [T][8]: cell_size, wall_height, agent_height, initial_life,
                                      max_life, step_reward, goal_reward, (pad) */
    /* Optional accelerator for SURVIVAL (NULL / 0 = sweep all n*n cells): per task, the cells that can ever
     * hold food — those with food_interval > 0 or food_rewards > 1e-2 — in ascending order. Every other cell's
     * wait flag, counter and food value never change (maze_base.py:83-88 only acts on cells whose counter
     * drops below 0 after a wait flag was set), so visiting just this list is exact. */
    const int16_t *food_cells;     /* [T][max_food] */
    const int32_t *n_food;         /* [T] */
    int32_t max_food;
    /* (ABI 5) The same list once more, laid out for the lane-per-env 2-D kernel whose neighbouring lanes run DIFFERENT tasks
     * (NULL = not provided; needed when mg_maze_state.food_by_slot is set): cell_slot inverts food_cells; slot_food /
     * slot_interval hold food_rewards / food_interval of the k-th listed cell of task t at [k * T + t], so lanes with consecutive
     * task ids (the default assignment e mod T) read consecutive addresses. */
    const int16_t *cell_slot;      /* [T][n*n]: index k of the cell in its task's list; for a cell that can never hold food -1 when
                                    * the task's food value there is exactly 0.0 (nothing is read for it), -2 when it is a nonzero
                                    * value <= 1e-2 (read from `food` for the 2-D observation) */
    const double *slot_food;       /* [max_food][T] */
    const int32_t *slot_interval;  /* [max_food][T] */
} mg_maze_tasks;

/* Per-env episode state (MazeBase.reset maze_base.py:40-63 + the 3-D cores). The SURVIVAL arrays
 * hold n*n cells per env; cell c of env e lives at index e*food_env_stride + c*food_cell_stride.
 * Use [N][n*n] (env_stride n*n, cell_stride 1) with the 3-D kernel (a workgroup per env reads a
 * contiguous row) and [n*n][N] (env_stride 1, cell_stride N) with the 2-D kernel (a lane per env).
 * They may be NULL for ESCAPE. */
typedef struct mg_maze_state {
    int32_t *task_id;     /* [N] index into the task table */
    int32_t *grid;        /* [2][N] _agent_grid */
    int32_t *steps;       /* [N] */
    int32_t *ori_idx;     /* [N] discrete-3D heading index 0..3 (maze_discrete_3d.py:46-48) */
    double *ori;          /* [N] continuous-3D heading, rad */
    float *loc;           /* [2][N] continuous-3D location (float32 like the reference array) */
    double *life;         /* [N] SURVIVAL */
    double *cur_food;     /* [N][n*n] SURVIVAL _cur_food_rewards (also the translucent-cell map) */
    uint8_t *wait_refresh;/* [N][n*n] SURVIVAL _food_wait_refresh (0/1) */
    int32_t *revival;     /* [N][n*n] SURVIVAL _food_revival_count */
    int64_t food_env_stride, food_cell_stride;   /* element strides of the three SURVIVAL arrays */
    /* (ABI 5) 1: the SURVIVAL arrays hold `max_food` food SLOTS per env instead of n*n cells — slot k (the k-th cell of the env's
     * task's food_cells list) of env e at e*food_env_stride + k*food_cell_stride; cells outside the list keep their task values
     * for ever and are not stored. For mg_maze2d_step / mg_maze_reset with [max_food][N] arrays (env_stride 1, cell_stride N): lane
     * e's k-th access is coalesced whatever task it runs — indexed by cell, every lane of a wave touched a different [n*n][N] row
     * (22x slower than ESCAPE at 2^20 envs). Needs mg_maze_tasks.food_cells / cell_slot / slot_food / slot_interval.
     * INVARIANT the slot path relies on: revival >= 0 wherever wait_refresh == 0 (true after every reset, renewal and step —
     * the counter only counts down while its slot waits); a slot that is not waiting is then left untouched without reading its
     * counter (maze_base.py:83-88 would renew a negative counter of a non-waiting cell: unreachable, and refused at load time). */
    int32_t food_by_slot;
} mg_maze_state;

/* First-person renderer constants (MazeCoreDiscrete3D.__init__ maze_discrete_3d.py:18-37 and the
 * call at :113-117) plus caller-prepared tables. */
typedef struct mg_maze_view {
    int32_t res_h, res_v;          /* resolution_horizon (image axis 0), resolution_vertical (axis 1) */
    double max_vision;             /* 12.0 */
    double l_focal;                /* 0.20 */
    double text_size;              /* 1.0 */
    double tan_half_fov;           /* numpy.tan(fol_angle / 2), fol_angle = 0.6 * 3.1415926 */
    double collision_dist;         /* 0.20 (continuous dynamics) */
    const double *col_cos;         /* DEVICE [res_h] cos_hp per screen column, see mg_maze_view_tables */
    const double *col_sin;         /* DEVICE [res_h] sin_hp */
    float ori_sin[4], ori_cos[4];  /* float32 sin/cos of the four discrete headings, computed by the
                                      caller with float32 numpy ufuncs like the reference */
    const uint32_t *textures;      /* DEVICE [n_textures][tex][tex] texels packed r | g<<8 | b<<16;
                                      texture 0 = ground, 1.. = walls (maze_task.py:19-36) */
    const uint32_t *ceil_texture;  /* DEVICE [tex][tex] */
    int32_t n_textures, tex_size;
    int32_t max_ray_records;       /* optional bound on translucent records per ray (0 = 2n+1). A ray crosses
                                      at most 2*floor(max_vision / min cell_size) + 4 cells before it stops.
                                      Hard limit 127 (the count travels in 7 bits between the two render passes):
                                      larger values, and 2n+1 > 127, are clamped to it */
    int32_t obs_format;            /* 0: int32 [N][res_h][res_v][3], the reference's dtype (values exceed 255);
                                      1: uint8 with saturation at 255 — a non-parity fast path (4x fewer HBM bytes) */
    double uniform_cell_size;      /* (ABI 5) > 0: the caller vouches that EVERY task of the table has exactly this cell_size (tasks
                                      of one sampler configuration do). The library then evaluates the renderer's power-of-two
                                      conditions once, on the host, and runs its specialised kernel when they all hold (cell size,
                                      texture size and resolution powers of two, int32 frames: the stock set-up) — same frames bit
                                      for bit. 0: unknown, the general kernel decides per env. (ABI 6) The promise is CHECKED, not trusted:
                                      see mg_maze_check_uniform_cell_size — a wrong value is MG_ERR_BAD_CONFIG, never wrong frames. */
} mg_maze_view;

/* (ABI 6) Check mg_maze_view.uniform_cell_size against a task table: reads the table's [T][8] scalar rows back to the host
 * (T * 64 bytes; SYNCHRONOUS on `stream` — a set_task-time call, not a step-time one) and compares every task's cell_size
 * (TaskConfig.cell_size, maze_task.py:15-17) with `uniform_cell_size`. MG_OK: the pair (tasks->scalars, value) is remembered
 * and mg_maze3d_step accepts it without looking again. MG_ERR_BAD_CONFIG: some task differs (mg_last_error names it).
 * mg_maze3d_step runs this check itself the first time it meets an unchecked (table, value) pair — one stream
 * synchronisation, once — and refuses an unchecked pair under stream capture (where it cannot synchronise). The memory of a
 * checked pair is keyed by the ADDRESS of the scalar rows: a caller that rewrites a checked table in place, or frees it and
 * uploads another one that lands at the same address, must call this again (an explicit call always re-reads; the Python layer
 * calls it at every set_task). `tasks->scalars` may be a host pointer (then it is read directly). */
int mg_maze_check_uniform_cell_size(const mg_maze_tasks *tasks, double uniform_cell_size, void *stream);

/* (ABI 7) Forget everything the library remembers about a task table (today: the checked (scalars address, value) pair above).
 * ALLOCATOR-REUSE HAZARD: the memory of a checked pair is keyed by address; a caching allocator (torch's, a pool) readily hands
 * the address of a freed table to the next one. A binding must call this BEFORE it frees a task table or rewrites its cell
 * sizes in place — the next mg_maze3d_step on that address then re-reads the rows (or refuses under capture) instead of
 * trusting a check made on other contents. The Python layer calls it from DeviceTaskTable's finaliser and re-checks at every
 * set_task. Host-only, never fails on an unknown table. */
int mg_maze_forget_tasks(const mg_maze_tasks *tasks);

/* Host helper: the per-column tables of ray_caster_utils.py:82-90 (tan_hp accumulated column by
 * column exactly like the reference loop). Writes res_h doubles to each HOST array; the caller
 * uploads them and points col_cos / col_sin at the device copies. */
int mg_maze_view_tables(int32_t res_h, double tan_half_fov, double l_focal, double *col_cos_host,
                        double *col_sin_host);

/* MazeBase.reset for the envs with mask[e] != 0 (NULL = all): agent to the task's start cell,
 * heading 0, steps 0, SURVIVAL food/life restored. */
int mg_maze_reset(const mg_maze_tasks *tasks, int32_t task_type, int32_t n_envs, const mg_maze_state *state,
                  const uint8_t *mask, void *stream);

/* On-device task generation — MazeTaskManager.sample_task (maze_task.py:41-190) for a whole task
 * table. Row t of every table array receives, bit for bit, the TaskConfig the reference returns after
 *     random.seed(seed_t); numpy.random.seed(seed_t); sample_task(n, allow_loops, ...)
 * (both MT19937 streams and numpy's pairwise float64 sum are reproduced on the device; the CPU
 * restatement is oracle/maze_sampler.py). seed_t = seeds[t] (DEVICE array) or seed_base + t when
 * seeds is NULL; seeds are 32-bit like numpy.random.seed's integer argument.
 * The outputs are the arrays an mg_maze_tasks table points at ([T][2], [T][n*n], [T][8]). */
typedef struct mg_maze_sample_params {
    int32_t n;               /* odd, 7..63 */
    int32_t allow_loops;
    int32_t n_texts;         /* MazeTaskManager.n_texts: ground + wall textures (7 for the shipped set) */
    int32_t food_interval;
    int32_t has_goal_reward; /* 0: goal_reward=None -> -sqrt(n)*n*step_reward (maze_task.py:163) */
    double cell_size, wall_height, agent_height;
    double step_reward, goal_reward, food_reward;
    double initial_life, max_life;
    double food_density, crowd_ratio;
} mg_maze_sample_params;

int mg_maze_sample_tasks(const mg_maze_sample_params *params, int32_t n_tasks, uint32_t seed_base,
                         const uint32_t *seeds, int32_t *start, int32_t *goal, int8_t *walls, uint8_t *texts,
                         double *food_rewards, int32_t *food_interval, double *scalars, void *stream);

/* MetaMaze2D.step (maze_env.py:189-204 -> maze_2d.py:21-34 + maze_base.py:65-95) and
 * update_observation (maze_2d.py:89-121).
 *   action i32 [N] in 0..3 (DISCRETE_ACTIONS maze_env.py:14); NULL = observe only (reset obs)
 *   obs f32 [N][2v+1][2v+1]; reward f32 [N] (may be NULL), reward64 f64 [N] (may be NULL), done u8 [N].
 * auto_reset != 0: an env whose step ended the episode is reset (same task) inside the launch and
 * its obs row is the first observation of the next episode. */
int mg_maze2d_step(const mg_maze_tasks *tasks, int32_t task_type, int32_t max_steps, int32_t view_grid,
                   int32_t auto_reset, int32_t n_envs, const mg_maze_state *state, const int32_t *action,
                   float *obs, float *reward, double *reward64, uint8_t *done, void *stream);

/* MetaMazeDiscrete3D.step (maze_env.py:59-75 -> maze_discrete_3d.py:51-81) or, with
 * continuous != 0, MetaMazeContinuous3D.step (maze_env.py:129-145 -> maze_continuous_3d.py:47-56 ->
 * dynamics.py:71-92), then evaluation_rule and the first-person render (ray_caster_utils.py:66-209)
 * with the SURVIVAL life bar (maze_discrete_3d.py:118-126).
 *   action: discrete i32 [N] in 0..3; continuous f32 [N][2] (turn, walk); NULL = observe only
 *   obs i32 [N][res_h][res_v][3] (values exceed 255, like the reference), or u8 with view->obs_format 1 */
int mg_maze3d_step(const mg_maze_tasks *tasks, const mg_maze_view *view, int32_t task_type, int32_t max_steps,
                   int32_t continuous, int32_t auto_reset, int32_t n_envs, const mg_maze_state *state,
                   const void *action, void *obs, float *reward, double *reward64, uint8_t *done,
                   void *stream);

/* ========================================================================================
 * MetaLocomotion walkers (humanoid / ant) — replaces, for N envs, WalkerBaseEnv.step
 * (metalocomotion/envs/utils/walker_base_env.py:43-82) including the physics the reference
 * delegates to pybullet.stepSimulation() (scene_bases.py:45-50).
 * PARITY UNPINNED for the physics: PyBullet is not part of the reference tree. The engine is a
 * from-scratch reduced-coordinate multibody solver (joint-space inertia matrix + Newton-Euler bias,
 * semi-implicit Euler, projected Gauss-Seidel contacts / joint limits) run with the reference's
 * parameters; see DESIGN.md §3.4 for the stated assumptions. The Python-side rules (torques,
 * observation, reward, done) follow the reference source exactly.
 * ======================================================================================== */

#define MG_WALKER_MAX_BODIES 16
#define MG_WALKER_MAX_JOINTS 24
#define MG_WALKER_MAX_SPHERES 128
#define MG_WALKER_MAX_FEET 6
#define MG_WALKER_MAX_GEOMS 24
#define MG_WALKER_MAX_PAIRS 128

/* Topology shared by every task of a batch (all MetaLocomotion variants of one robot share it). */
typedef struct mg_walker_topology {
    int32_t n_bodies, n_joints, n_spheres, n_feet;
    int32_t body_parent[MG_WALKER_MAX_BODIES];     /* -1 for the floating base (body 0); a parent comes before its children
                                                      (body_parent[b] < b), else MG_ERR_BAD_CONFIG */
    int32_t joint_body[MG_WALKER_MAX_JOINTS];      /* non-decreasing; joints of a body act in order */
    int32_t sphere_body[MG_WALKER_MAX_SPHERES];    /* collision spheres (capsule end caps, sphere geoms) */
    int32_t foot_body[MG_WALKER_MAX_FEET];         /* bodies whose ground contact sets feet_contact */
    /* self-collision (robot_bases.py:119: URDF_USE_SELF_COLLISION | ..._EXCLUDE_ALL_PARENTS): capsule
     * geoms and the geom pairs to test (bodies distinct, not ancestor-related, not welded together) */
    int32_t n_geoms, n_pairs;
    int32_t geom_body[MG_WALKER_MAX_GEOMS];
    uint8_t pair_a[MG_WALKER_MAX_PAIRS], pair_b[MG_WALKER_MAX_PAIRS];
    /* Which feet_contact flag a collision proxy reports to: f in [0, n_feet) or -1 (not a foot). A MetaLocomotion foot is a
     * whole body (walker_base_env.py:57-63: sphere_foot[g] = f iff sphere_body[g] == foot_body[f]); a URDF robot whose fixed
     * links were merged into their parents keeps the LINK a proxy came from this way (the A1's toe spheres on the calf body:
     * quadrupedal/robots/a1.py:299-312 GetFootContacts looks at the toe links only). Proxies with -1 that touch the ground or
     * the terrain are counted in mg_walker_state.bad_contacts (a1.py:314-323 GetBadFootContacts). */
    int8_t sphere_foot[MG_WALKER_MAX_SPHERES];
} mg_walker_topology;

/* Per-task geometry / inertia table, doubles, one row of `model_stride` values per task:
 *   body_pos[nb][3] body_rot[nb][9] body_mass[nb] body_com[nb][3] body_inertia[nb][9]
 *   joint_anchor[nj][3] joint_axis[nj][3] joint_lo[nj] joint_hi[nj] joint_armature[nj]
 *   joint_damping[nj] joint_stiffness[nj] motor_torque[nj] sphere_pos[ns][3] sphere_radius[ns]
 *   geom_p0[ng][3] geom_p1[ng][3] geom_radius[ng]      (capsule end points in the body frame)
 *   [sphere_margin[ns]]                                 (ABI 7, only with mg_walker_params.sphere_margin_in_table: per-proxy contact margins)
 * (motor_torque[j] = motor_power_j * power, the factor multiplying clip(a_j,-1,1): humanoids.py:50-54,
 * walker_base.py:26-29). */
typedef struct mg_walker_models {
    const double *table;       /* DEVICE [n_tasks][model_stride] */
    int32_t n_tasks, model_stride;
} mg_walker_models;

#define MG_WALKER_BOX_DOUBLES 16

typedef struct mg_walker_params {
    double time_step;          /* 0.005  walker_base_env.py:7 */
    int32_t frame_skip;        /* 4      sub-steps per env step */
    int32_t solver_iterations; /* 5      scene_bases.py:17 */
    double erp;                /* 0.9    contact ERP, scene_bases.py:55 */
    double limit_erp;          /* 0.2    joint-limit ERP (Bullet's default constraint ERP) */
    double gravity;            /* 9.8    env_bases.py:48 */
    double friction;           /* 0.64 = ground 0.8 (stadium.py:23) x geom 0.8 (humanoid.xml:5) */
    double alive_z, alive_bonus, dead_bonus;   /* humanoids.py:56: +2 if z > 0.50 else -1 */
    double initial_z;          /* humanoids.py:48: 0.8 */
    double joints_at_limit_cost;   /* -0.1 walker_base_env.py:22 */
    double walk_target_x, walk_target_y;       /* 1e3, 0 */
    int32_t max_steps;
    int32_t floor_in_parts;    /* 1: the floor link counts in the mean part position (walker_base_env.py:30-31): true
                                  from the first step on and for every later reset of the same robot object; 0 for
                                  the FIRST reset after a set_task (the floor joins robot.parts after robot.reset()).
                                  The mean runs over all of robot.parts: the base once, every other body once per hinge
                                  joint it carries (min. 1) */
    int32_t mapping;           /* 1 (default): wave per env, LDS-resident; 0: lane per env (cross-check) */
    int32_t self_collision;    /* 1: capsule-capsule contacts between the topology's geom pairs */
    double self_friction;      /* geom friction squared (Bullet multiplies the two coefficients) */
    /* Fused auto-reset (not in the reference: replaces the user's `if done: env.reset()` round trip).
     * An env whose step ended the episode is reset inside the launch — robot_specific_reset
     * (walker_base.py:13-24): base pose from the model, every joint at U(-0.1, 0.1) with zero velocity —
     * and its obs row is the first observation of the next episode. Joint noise comes from
     * Philox4x32-10 keyed by `seed`, counter (env_id_base + env, step_index, joint/4), so it does not
     * depend on how envs are sharded. The caller advances step_index by one per launch. */
    int32_t auto_reset;
    uint64_t seed, step_index, env_id_base;
    /* The reference's own float choreography (tests/golden/walker_rules.npz, recorded from the unmodified Python):
     * torque_f32 1 = Humanoid.apply_action humanoids.py:50-54 (python floats times a float32 action: float32 product),
     *            0 = WalkerBase.apply_action walker_base.py:26-29 (float() first: float64 product) — the ant;
     * height_f32 1 = the alive test adds the python float initial_z to the float32 obs[0] in float32 (humanoid),
     *            0 = initial_z is a float64 taken from the first calc_state (ant). */
    int32_t torque_f32, height_f32;
    /* Actuators evaluated INSIDE the launch, once per physics sub-step (wave mapping only; 0 keeps the action path above):
     *   1  position control: torque_j = LaikagoMotorModel.convert_to_torque (quadrupedal/robots/laikago_motor.py:136-168,
     *      the same expression as mg_a1_apply_action) of the CURRENT joint state (pd latency 0, the A1 default) and the
     *      desired angles in pd_command — what Minitaur.Step does 13 times per env step around stepSimulation;
     *   2  raw torques from pd_command (the caller ran the motor model itself);
     *   3  (shape-generic kernels) HYBRID commands, laikago_motor.py:143-153: pd_command is [5 nj][N], row 5 j + k of motor j =
     *      desired angle, kp, desired rate, kd, additional torque; strength and torque limit as in mode 1;
     *   4  TORQUE mode, laikago_motor.py:125-128: torque_j = strength_j * pd_command_j, no limit.
     * pd_kp_env / pd_kd_env (shape-generic kernels; DEVICE f64 [nj][N] or NULL): per-robot gains for mode 1, what
     * locomotion_gym_env.py:388-392 draws when the dynamics are randomised.
     * pd_command: DEVICE f64 [nj][N] ([5 nj][N] in mode 3). substep_log: DEVICE f64 [frame_skip][3 nj + 7][N] or NULL — after every sub-step the
     * joint angles, rates, applied torques, the base quaternion (x y z w) and the body-frame angular velocity, i.e. one
     * Minitaur.GetTrueObservation (minitaur.py:1175-1182) per sub-step, ready for mg_a1_receive_log. */
    int32_t actuation;
    const double *pd_command;
    double pd_kp[MG_WALKER_MAX_JOINTS], pd_kd[MG_WALKER_MAX_JOINTS], pd_strength[MG_WALKER_MAX_JOINTS],
        pd_limit[MG_WALKER_MAX_JOINTS];
    double *substep_log;
    /* Static terrain on top of the ground plane (wave mapping only): n_terrain_boxes oriented boxes shared by every env —
     * what the quadrupedal tasks build in their Bullet world (quadrupedal/envs/utilities/terrain.py; the box lists come from
     * metagym_amd/quadrupedal/terrain.py). terrain: DEVICE f64 [n_terrain_boxes][MG_WALKER_BOX_DOUBLES]: position[3],
     * R[9] (row-major, box -> world), half extents[3], mu (the contact's friction coefficient: Bullet multiplies the two
     * bodies' lateral frictions). Per collision sphere the deepest penetrated box (first on ties) gives one contact: normal
     * from the closest surface point to the sphere centre, or — centre inside the box — the face of least penetration. */
    int32_t n_terrain_boxes;
    const double *terrain;
    /* Per-proxy lateral friction (shape-generic wave kernels only; NULL = one coefficient for the whole robot, above):
     * DEVICE f64 [n_spheres], the coefficient of the LINK each collision proxy belongs to. Bullet multiplies the two bodies'
     * coefficients, so with this set `friction` is the ground plane's OWN coefficient and terrain[.][15] each box's own
     * (quadrupedal: plane 5, locomotion_gym_env.py:258; boxes 5, terrain.py:14; feet SetFootFriction(1), :408). */
    const double *sphere_friction;
    /* Velocity damping of every body, btMultiBody's m_linearDamping / m_angularDamping (default 0.04 each, what
     * pybullet.changeDynamics documents): force -m v (k + k |v|) at the centre of mass, torque -I w (k + k |w|). 0 = off.
     * The quadrupedal reference switches it off (minitaur.py:346-353 at :419); MetaLocomotion never touches it, so PyBullet's
     * default applies there — the `preset="bullet"` default of metalocomotion.mjcf / WalkerBatchEnv (the two tuned wave kernels
     * have a damped instantiation each; the lane mapping carries it too). */
    double body_linear_damping, body_angular_damping;
    const double *pd_kp_env, *pd_kd_env;
    /* External push on the base body during the FIRST sub-step of the launch only (shape-generic kernels; NULL = none):
     * DEVICE f64 [6][N] — force (3) and application point (3), both in the base BODY frame. What
     * pybullet.applyExternalForce(body, -1, force, pos, LINK_FRAME) followed by 13 stepSimulation() calls does: Bullet clears
     * external forces after every stepSimulation (RandomWrapper, quadrupedal/envs/env_wrappers/MonitorEnv.py:530-535,644-660;
     * the caller adds the root link's inertial offset to pos: PyBullet's link frame is the inertial frame). */
    const double *ext_wrench;
    /* (ABI 5) btMultiBody's m_maxCoordinateVelocity (default 100, never changed by the reference): at the end of every sub-step
     * each of the 6 + nj generalized velocities is clamped to [-v, v] before the positions are integrated — the
     * processDeltaVeeMultiDof2 -> applyDeltaVeeMultiDof application; Bullet clamps the unconstrained velocities the same way,
     * which is not restated. 0 = off (the `preset="mujoco"` world). All three mappings. */
    double max_coordinate_velocity;
    /* (ABI 5) Per-robot terrains: with terrain_id != NULL, `terrain` is a TABLE of n_terrain_tables courses of n_terrain_boxes
     * boxes each — DEVICE f64 [n_terrain_tables][n_terrain_boxes][MG_WALKER_BOX_DOUBLES], shorter courses padded with boxes of zero
     * half extents parked far away (x = 1e30) — and robot e stands on course terrain_id[e] (DEVICE i32 [N], read at launch
     * time: a masked reset may rewrite entries to move robots to another course, the maze task-table pattern). What
     * LocomotionGymEnv.reset(hardset=True, mode=..., ...) does per episode for ONE robot (quadrupedal/envs/
     * locomotion_gym_env.py:297-301): a new terrain task per episode. NULL: one course for the whole batch, as before. */
    const int32_t *terrain_id;
    int32_t n_terrain_tables;
    /* (ABI 5) Per-robot dynamics, what LocomotionGymEnv.reset redraws for its one robot when random_dynamic is set
     * (quadrupedal/envs/locomotion_gym_env.py:381-413) — shape-generic wave kernels, NULL = the shared values above:
     *   gravity_env        DEVICE f64 [3][N]: the world's gravity ACCELERATION vector for robot e (pybullet.setGravity(gx, gy, gz)
     *                      :407; the scalar `gravity` above is (0, 0, -gravity)). The reference draws gz from U(8, 12) — positive,
     *                      i.e. pointing up — and hands it to setGravity as it is.
     *   foot_friction_env  DEVICE f64 [N]: the lateral friction of robot e's FOOT proxies (those with sphere_foot >= 0), replacing
     *                      their sphere_friction entry (Minitaur.SetFootFriction :408); needs sphere_friction.
     * Per-robot masses and inertias need no field: give every robot its own row of the model table (task_id[e] = e). */
    const double *gravity_env;
    const double *foot_friction_env;
    /* (ABI 5) mg_walker_reset only: where a reset places the base body instead of the model's own start pose — DEVICE f64
     * reset_pos [3][N] (world position of the base body's origin) and reset_rot [9][N] (row-major rotation), each NULL = the model's.
     * Minitaur.Reset(default_pose=, yaw=) (quadrupedal/robots/minitaur.py:425-431, envs/locomotion_gym_env.py:334-338). With either given, a
     * reset also zeroes the reset robots' bad_contacts / foot_force entries (no contact points yet). */
    const double *reset_pos;
    const double *reset_rot;
    /* (ABI 7) Bullet's contact-breaking margin, one length for every proxy (0 = penetration only, the behaviour up to ABI 6; see
     * sphere_margin_in_table below for Bullet's own per-link rule).
     * A collision proxy whose surface is within its margin ABOVE the ground plane or a terrain box is a contact point, as in a
     * Bullet manifold: (i) it gets a normal row whose bias is erp * depth / time_step while it penetrates (depth >= 0) and the
     * SPECULATIVE depth / time_step (< 0, no ERP) while it is separated — btMultiBodyConstraintSolver::setupMultiBodyContactConstraint:
     * `penetration = distance + slop > 0` -> velocityError -= penetration / dt — so the proxy may close at most its gap per
     * sub-step and a resting contact keeps its rows instead of flickering; the two friction rows are bounded by mu x that
     * normal multiplier as always; (ii) feet_contact / bad_contacts count EVERY proxy inside the margin — what
     * getContactPoints returns (walker_base_env.py:57-63 via robot_bases.py:291-292) — whether or not the solver's cap kept it.
     * Self-collision pairs stay penetration-only. The cap (all mappings, any margin): candidates are collected in candidate order
     * (ground per proxy, terrain per proxy, self pairs; at most 48), and when more than 12 exist the 12 DEEPEST are kept (ties:
     * the earlier candidate), in candidate order — penetrating points before speculative ones. */
    double contact_margin;
    /* (ABI 7) Per-proxy contact margins: 1 = every row of the model table carries n_spheres margins (metres) BEHIND geom_radius
     * (model_stride >= 25 nb + 12 nj + 5 ns + 7 ng then) and they replace contact_margin; 0 = contact_margin for every proxy.
     * Bullet's margin is RELATIVE by default — btCollisionDispatcher is constructed with
     * CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD, so a manifold breaks at min over the two shapes of
     * gContactBreakingThreshold (0.02) x btCollisionShape::getAngularMotionDisc(): 2 % of the LINK's size (bounding-sphere radius
     * of its compound shape's AABB + the distance of the AABB's centre from the shape's origin), not 2 cm: 3 - 8 mm for the
     * humanoid's links, 0.7 mm for the A1's 2 cm toe spheres. metagym_amd.metalocomotion.mjcf.contact_margins(model, "relative")
     * computes a row's entries. */
    int32_t sphere_margin_in_table;
} mg_walker_params;

/* Per-env state, SoA doubles: component c of env e at base[c*N + e]. */
typedef struct mg_walker_state {
    int32_t *task_id;     /* [N] */
    double *pos;          /* [3][N] base body origin (world) */
    double *rot;          /* [9][N] base body orientation (row-major) */
    double *vel;          /* [3][N] base origin velocity (world) */
    double *omega;        /* [3][N] base angular velocity (world) */
    double *q, *qd;       /* [nj][N] */
    double *potential;    /* [N] */
    float *feet_contact;  /* [nf][N] */
    int32_t *steps;       /* [N] */
    int32_t *bad_contacts; /* [N] or NULL: ground / terrain contact points of the last sub-step on proxies that are no foot */
    double *foot_force;    /* [nf][N] or NULL (shape-generic wave kernels): |sum of normal impulse x contact normal| / time_step over
                              each foot's ground / terrain contact points in the last sub-step, newtons — what a1.py:325-356
                              GetFootContactsForce adds up from PyBullet's contact points (SimpleFootForceSensor) */
} mg_walker_state;

/* WalkerBaseEnv.reset: base to its model pose, joints to joint_noise (f64 [nj][N], the caller draws
 * U(-0.1,0.1) like walker_base.py:15; NULL = zeros), velocities zero, for envs with mask != 0
 * (NULL = all); writes the reset observation rows (obs may be NULL). */
int mg_walker_reset(const mg_walker_topology *topo, const mg_walker_models *models, const mg_walker_params *prm,
                    int32_t n_envs, const mg_walker_state *state, const uint8_t *mask, const double *joint_noise,
                    float *obs, void *stream);

/* WalkerBaseEnv.step for all envs: torques from action (f32 [N][nj]; NULL when prm->actuation != 0: the in-launch actuators
 * read prm->pd_command instead), frame_skip physics sub-steps,
 * calc_state -> obs f32 [N][8 + 2 nj + nf], reward f32 [N], rewards5 f32 [N][5] (alive, progress,
 * electricity, joints_at_limit, feet_collision; may be NULL), done u8 [N]. */
int mg_walker_step(const mg_walker_topology *topo, const mg_walker_models *models, const mg_walker_params *prm,
                   int32_t n_envs, const mg_walker_state *state, const float *action, float *obs, float *reward,
                   float *rewards5, uint8_t *done, void *stream);

/* ========================================================================================
 * Quadrupedal (Unitree A1) — the ACTUATION path of metagym/quadrupedal/robots/minitaur.py + a1.py +
 * laikago_motor.py for N robots: everything `Minitaur._StepInternal` (minitaur.py:232-238) does on either side of
 * `pybullet.stepSimulation()`. The A1 body itself is NOT here: a1/a1.urdf ships with pybullet_data and the physics is
 * PyBullet — neither is in the reference tree. Pinned bit for bit to the unmodified
 * reference by tests/golden/a1_actuation.npz (oracle/gen_golden_a1.py).
 *
 * One sub-step of the reference is   ApplyAction -> stepSimulation -> ReceiveObservation:
 *   mg_a1_apply_action          A1.ApplyAction a1.py:451-483 (optional command clip), Minitaur.ApplyAction
 *                               minitaur.py:906-955, ProcessAction :1419-1436 (action interpolation),
 *                               _GetPDObservation / _GetDelayedObservation :1205-1232 (pd latency),
 *                               LaikagoMotorModel.convert_to_torque laikago_motor.py:92-169
 *   (the caller advances its physics with the returned torques)
 *   mg_a1_receive_observation   ReceiveObservation minitaur.py:1184-1203: GetTrueObservation :1175-1182 pushed on the
 *                               history deque (maxlen 100, :139), control observation = history delayed by the
 *                               control latency
 *   mg_a1_sensors               GetMotorAngles / Velocities / Torques :755-810, GetBaseRollPitchYawRate :874-885,
 *                               GetEnergyConsumptionPerControlStep :812-820 (sensor noise is zero, :48)
 * All values are float64 like the reference's numpy arrays. Arrays are SoA: component c of robot e at base[c*N + e].
 * ======================================================================================== */

#define MG_A1_NUM_MOTORS 12
#define MG_A1_OBS_DIM 43          /* motor angles 12, velocities 12, torques 12, base quaternion 4, rpy rate 3 */
enum { MG_A1_MODE_POSITION = 1, MG_A1_MODE_TORQUE = 2, MG_A1_MODE_HYBRID = 3 };   /* robot_config.py:13-27 */

typedef struct mg_a1_actuator_config {
    double time_step;              /* 0.002  locomotion_gym_config.py:18 */
    int32_t action_repeat;         /* 13     env_builder.py:45 */
    int32_t history_len;           /* 100    minitaur.py:139 (deque maxlen); any value > latency / time_step + 1 gives
                                      the same results once that many observations exist */
    int32_t mode;                  /* MG_A1_MODE_* */
    int32_t clip_commands;         /* A1._ClipMotorCommands a1.py:465-483 (POSITION commands only) */
    double max_angle_change;       /* 0.2    a1.py:52 */
    double control_latency, pd_latency;               /* seconds; used when the per-robot arrays below are NULL */
    const double *control_latency_env, *pd_latency_env;   /* DEVICE [N] or NULL (locomotion_gym_env.py:349,374-375) */
    double kp[MG_A1_NUM_MOTORS], kd[MG_A1_NUM_MOTORS];    /* a1.py:63-68 */
    const double *kp_env, *kd_env;                        /* DEVICE [12][N] or NULL (SetMotorGains, :388-392) */
    double strength[MG_A1_NUM_MOTORS];                    /* laikago_motor.py:58 */
    double torque_limit[MG_A1_NUM_MOTORS];                /* 33.5 minitaur.py:88 */
    int32_t has_torque_limit;
} mg_a1_actuator_config;

typedef struct mg_a1_actuator_state {
    double *history;          /* DEVICE [history_len][43][N] ring of true observations */
    int32_t *count;           /* DEVICE [N] observations held (<= history_len) */
    int32_t *head;            /* DEVICE [N] ring slot of the newest observation */
    double *observed_torque;  /* DEVICE [12][N] torques of the last ApplyAction (enter the next observation) */
    double *control_obs;      /* DEVICE [43][N] observation delayed by the control latency */
} mg_a1_actuator_state;

/* command: DEVICE f64 [12][N] (POSITION / TORQUE) or [60][N] (HYBRID, laikago_motor.py:143-153). last_command may be
 * NULL; otherwise the command used is last + lerp * (command - last) (ProcessAction, lerp = (substep + 1) / repeat).
 * torque: DEVICE f64 [12][N] out (what _SetMotorTorqueByIds hands to the physics). */
int mg_a1_apply_action(const mg_a1_actuator_config *cfg, int32_t n_envs, const mg_a1_actuator_state *state,
                       const double *command, const double *last_command, double lerp, double *torque, void *stream);
/* q, qd: DEVICE f64 [12][N] true motor angles / rates; base_quat [4][N] (x y z w, relative to the initial
 * orientation); rpy_rate [3][N] angular velocity in the body frame. clear_mask: DEVICE u8 [N] or NULL — per robot 0 = push,
 * 1 = empty the history first (Minitaur.Reset, minitaur.py:437), 2 = leave this robot untouched (the first observation
 * after a reset of PART of the batch: the others are between two sub-steps and must not see a second push). */
int mg_a1_receive_observation(const mg_a1_actuator_config *cfg, int32_t n_envs, const mg_a1_actuator_state *state,
                              const double *q, const double *qd, const double *base_quat, const double *rpy_rate,
                              const uint8_t *clear_mask, void *stream);
/* mg_a1_receive_observation immediately followed by mg_a1_apply_action of the NEXT sub-step (they are adjacent in
 * Minitaur.Step's loop: ... stepSimulation, ReceiveObservation | ApplyAction, stepSimulation ...) as ONE launch: the
 * observation just pushed is the one the PD term reads (pd latency 0), so it never travels back from HBM. Same arguments as
 * the two calls; results are bit-identical to calling them one after the other. */
int mg_a1_receive_and_apply(const mg_a1_actuator_config *cfg, int32_t n_envs, const mg_a1_actuator_state *state,
                            const double *q, const double *qd, const double *base_quat, const double *rpy_rate,
                            const double *command, const double *last_command, double lerp, double *torque, void *stream);
/* K ReceiveObservation calls at once from a sub-step log (mg_walker_params.substep_log, [K][43][N]: q 12, qd 12, torque 12,
 * quaternion 4, body rate 3 per sub-step): the K observations are pushed on the history in order and the control observation
 * is refreshed once at the end (only the last one is ever read between env steps). The logged torques become the observed
 * torques. Identical to K mg_a1_receive_observation calls each preceded by the ApplyAction that produced that torque. */
int mg_a1_receive_log(const mg_a1_actuator_config *cfg, int32_t n_envs, const mg_a1_actuator_state *state, const double *log,
                      int32_t n_substeps, void *stream);
/* Any output may be NULL. motor_angles / motor_velocities / motor_torques: f64 [12][N]; rpy_rate f64 [3][N];
 * energy f64 [N]. */
int mg_a1_sensors(const mg_a1_actuator_config *cfg, int32_t n_envs, const mg_a1_actuator_state *state,
                  double *motor_angles, double *motor_velocities, double *motor_torques, double *rpy_rate,
                  double *energy, void *stream);

/* ---- Control-side wrappers of A1GymEnv.step (envs/env_wrappers/MonitorEnv.py), pinned by tests/golden/a1_control.npz ----
 * Transcendentals (sin, exp, arccos, arcsin, arctan2, tanh) come from the device math library: results agree with the
 * reference's libm to a few ulp (tests use 1e-12), everything else is the reference's float64 arithmetic in its order. */

#define MG_A1_ETG_MAX_H 32

/* ETGWrapper (MonitorEnv.py:222-273): ETG_layer.update2 + ETG_model.forward + act_clip
 * (envs/utilities/ETG_model.py:38-55,98-130; leg IK robots/a1.py:88-102,493-524), then
 * TrajectoryGeneratorWrapperEnv.step -> LaikagoPoseOffsetGenerator.get_action (simple_openloop.py:144-165). */
typedef struct mg_a1_etg_config {
    int32_t enabled;               /* ETG != 0; 0: command = generator(action) only */
    int32_t H;                     /* 20   number of radial basis functions (<= MG_A1_ETG_MAX_H) */
    double T, T2_ratio;            /* 0.5, 0.5 */
    double sigma_sq, amp;          /* 0.04, 0.2  MonitorEnv.py:238 */
    double phase[2];               /* (-pi/2, 0) MonitorEnv.py:236 */
    double omega;                  /* 2 pi / T   ETG_model.py:20 (host value, so it is numpy's) */
    double u[MG_A1_ETG_MAX_H][2];  /* RBF centres, ETG_model.py:22-25 (host: numpy's sin) */
    double w[3][MG_A1_ETG_MAX_H], b[3];   /* ETG_w, ETG_b (the evolved parameters, MonitorEnv.py:240-245) */
    int32_t act_mode_pose;         /* 1: act_mode "pose" (tanh scaling); 0: "traj" (foot trajectory + IK) */
    int32_t gallop;                /* task_mode == "gallop" leg assignment, ETG_model.py:106-115 */
    double etg_weight;             /* 1    MonitorEnv.py:239 */
    int32_t action_space;          /* LaikagoPoseOffsetGenerator action_mode 0..3 */
    double pose[MG_A1_NUM_MOTORS]; /* (0, 0.9, -1.8) x 4  laikago_pose_utils.py:17-19 */
} mg_a1_etg_config;

/* last_etg_act: DEVICE f64 [12][N] state (ETGWrapper.last_ETG_act). t: DEVICE f64 [N], time since reset BEFORE this
 * env step (locomotion_gym_env get_time_since_reset). action: DEVICE f64 [12][N], or NULL = ETGWrapper.reset (only the
 * ETG state is refreshed, no command). command: DEVICE f64 [12][N] out — what reaches LocomotionGymEnv.step, i.e. the
 * input of mg_a1_apply_action. etg_obs: DEVICE f64 [H][N] out or NULL (info["ETG_obs"]). */
int mg_a1_etg_action(const mg_a1_etg_config *cfg, int32_t n_envs, double *last_etg_act, const double *action,
                     const double *t, double *command, double *etg_obs, void *stream);

#define MG_A1_MAX_SEGMENTS 32   /* the reference task terrains report up to 26 stretches (stairslope, slopeslope) */

/* RewardShaping (MonitorEnv.py:275-519). */
typedef struct mg_a1_reward_config {
    double w_torso, w_up, w_feet, w_tau, w_badfoot, w_footcontact;   /* Param_Dict MonitorEnv.py:12 */
    double reward_p, vel_d;        /* 1.0, 0.6 */
    double cw_half, cw_04;         /* arctanh(sqrt(0.95)) / 0.5 and / 0.4: c_prec's w (:421-425), host (numpy) values */
    int32_t n_segments;            /* info["env_info"] rows (locomotion_gym_env.py:76): x0, x1, upslope, downslope, angle */
    double seg[MG_A1_MAX_SEGMENTS][5];
    int32_t vel_mode;              /* 0 "max": min(vel_d, v) (the default); 1 "equal": exp(-5 |v - vel_d|)  (MonitorEnv.py:512-518) */
    /* (ABI 5) Per-robot terrains — the reference rebuilds its terrain, and with it info["env_info"], in reset(hardset=True, ...)
     * (locomotion_gym_env.py:297-301): with terrain_id != NULL robot e's stretches are rows [0, seg_count[t]) of
     * seg_table[t], t = terrain_id[e] (the same index mg_walker_params.terrain_id selects the boxes with), and n_segments /
     * seg above are not read. */
    const double *seg_table;       /* DEVICE f64 [T][MG_A1_MAX_SEGMENTS][5] */
    const int32_t *seg_count;      /* DEVICE i32 [T] */
    const int32_t *terrain_id;     /* DEVICE i32 [N] */
} mg_a1_reward_config;

typedef struct mg_a1_reward_state {
    double *last_base;     /* DEVICE [3][N]  last_basepose */
    double *last_base10;   /* DEVICE [30][N] last_base10 (10 x 3, newest first) */
    double *last_foot;     /* DEVICE [12][N] last_footposition (world frame, 4 x 3) */
    double *vd2;           /* DEVICE [2][N]  third component of the mutable default `vd` of re_torso / re_feet (:475,:430):
                              written on slopes only and never cleared, so it outlives the step (and reset()) */
    int32_t *steps;        /* DEVICE [N] */
} mg_a1_reward_state;

/* RewardShaping.reset (:305-318): base [3][N], rot_mat [9][N], footposition (base frame) [12][N] of the RESET info;
 * mask u8 [N] or NULL = all. */
int mg_a1_reward_reset(const mg_a1_reward_config *cfg, int32_t n_envs, const mg_a1_reward_state *state, const double *base,
                       const double *rot_mat, const double *footposition, const uint8_t *mask, void *stream);
/* RewardShaping.step (:320-366) on this step's info: base [3][N], pose (roll, pitch, yaw) [3][N], rot_mat [9][N],
 * footposition [12][N], real_contact f64 [4][N] (0/1), energy [N], bad_contacts i32 [N], d_yaw [N] or NULL (= 0).
 * Out: terms f64 [6][N] (torso, up, feet, tau, badfoot, footcontact; may be NULL), reward f64 [N], done u8 [N]. */
int mg_a1_reward_step(const mg_a1_reward_config *cfg, int32_t n_envs, const mg_a1_reward_state *state, const double *base,
                      const double *pose, const double *rot_mat, const double *footposition, const double *real_contact,
                      const double *energy, const int32_t *bad_contacts, const double *d_yaw, double *terms,
                      double *reward, uint8_t *done, void *stream);

/* The sensor stack behind A1GymEnv's observation (envs/env_builder.py:62-80, SENSOR_MODE dis / imu / motor / contact = 1):
 * BaseDisplacementSensor(convert_to_local_frame) robot_sensors.py:217-312, IMUSensor(R P Y dR dP dY) :314-437,
 * MotorAngleAccSensor :85-162, FootContactSensor :552-578, ordered by sensor name (locomotion_gym_env.py:621-632) and
 * flattened (env_utils.py:11-42): obs[0:3] base displacement, [3:7] foot contacts, [7:13] IMU, [13:37] motor angles and
 * their finite-difference rates. Pinned by tests/golden/a1_sensors.npz. */
#define MG_A1_SENSOR_OBS_DIM 37
typedef struct mg_a1_sensor_config {
    int32_t normal;            /* 1: (x - mean) / std of each sensor (robot_sensors.py:117-118,261-262,347-348) */
    double disp_dt;            /* 0.026  BaseDisplacementSensor's own default (:225) */
    double motor_dt;           /* num_action_repeat * sim_time_step (env_builder.py:49,73) */
} mg_a1_sensor_config;
typedef struct mg_a1_sensor_state {
    double *base_last, *base_cur;   /* DEVICE [3][N] */
    double *yaw;                    /* DEVICE [2][N] last, current */
    double *first_rpy;              /* DEVICE [3][N] */
    double *last_angle;             /* DEVICE [12][N] */
    int32_t *first;                 /* DEVICE [N] bit 0: IMU first_time, bit 1: MotorAngleAcc first_time */
    /* (ABI 5) sensor_mode["noise"] (env_builder.py:60-71): this observation's Gaussian draws, DEVICE f64 [33][N], already scaled by
     * their sigma, or NULL = no noise. Slots: displacement dx dy dz (sigma 1e-2, added BEFORE the rotation into the local frame,
     * robot_sensors.py:281-284), rpy 3 (6e-2) and drpy 3 (1e-1) (:399-402), motor angles 12 (1e-2) and rates 12 (0.5) — added
     * AFTER the rate was formed, and the noisy angles become last_angle (:146-149). The caller draws them (any generator);
     * pinned by tests/golden/a1_sensors_noise.npz with the reference's own draws as inputs. */
    const double *noise;
} mg_a1_sensor_state;
/* One observation per robot. reset_mask (u8 [N] or NULL = none; 2 = skip this robot, its sensor state and obs row stay): robots that were just reset — sensor.reset() + on_reset
 * (locomotion_gym_env.py:231-232,426-427) instead of on_step (:521-522). base [3][N] (GetBasePosition), rpy [3][N]
 * (GetBaseRollPitchYaw), drpy [3][N], motor_angles [12][N] (mg_a1_sensors), contact [4][N] (0 / 1). obs: f64 [N][37]. */
int mg_a1_observation(const mg_a1_sensor_config *cfg, int32_t n_envs, const mg_a1_sensor_state *state, const double *base,
                      const double *rpy, const double *drpy, const double *motor_angles, const double *contact,
                      const uint8_t *reset_mask, double *obs, void *stream);

/* ObservationWrapper (envs/env_wrappers/MonitorEnv.py:77-221): the entries it appends to the sensor observation, in its
 * order. flags: MG_A1_EXTRA_ETG = info["ETG_act"] (12; (x - ETG_mean) / ETG_std of :89-94 when `normal`),
 * MG_A1_EXTRA_ETG_OBS = info["ETG_obs"] (etg_h), MG_A1_EXTRA_YAW = cos / sin(d_yaw - yaw) (:204-211; d_yaw [N] or NULL = 0).
 * etg_act: DEVICE f64 [12][N] (ETGWrapper.last_ETG_act), etg_obs: [etg_h][N], pose: [3][N] (roll, pitch, yaw).
 * out: DEVICE f64 [N][width], width = 12 * ETG + etg_h * ETG_OBS + 2 * YAW. (force_vec / dynamic_vec come from the
 * caller's physics and the RNN stacking is a copy: neither needs a kernel.) */
#define MG_A1_EXTRA_ETG 1
#define MG_A1_EXTRA_ETG_OBS 2
#define MG_A1_EXTRA_YAW 4
int mg_a1_observation_extras(int32_t n_envs, int32_t flags, int32_t normal, int32_t etg_h, const double *etg_act,
                             const double *etg_obs, const double *pose, const double *d_yaw, double *out, void *stream);

/* ActionFilter.filter / init_history / reset (quadrupedal/robots/action_filter.py:70-99), as Minitaur._FilterAction uses it
 * on the policy's motor commands (minitaur.py:1438-1457): per joint
 *   y = x b0 + sum_k xhist[k] b[k+1] - sum_k yhist[k] a[k+1],   history depth = order (low-pass) or 2 order (band-pass).
 * Coefficients come from the host (scipy.signal.butter like action_filter.py:160-185). Pinned by tests/golden/a1_filter.npz. */
#define MG_A1_FILTER_MAX_HIST 4
typedef struct mg_a1_filter_config {
    int32_t hist_len;                                             /* 1..4 */
    double a[MG_A1_NUM_MOTORS][MG_A1_FILTER_MAX_HIST + 1];        /* normalised: a[j][0] == 1 */
    double b[MG_A1_NUM_MOTORS][MG_A1_FILTER_MAX_HIST + 1];
} mg_a1_filter_config;
/* xhist, yhist: DEVICE f64 [hist_len][12][N] state (newest first). x: DEVICE f64 [12][N]. y: DEVICE f64 [12][N] out (may alias
 * x). init_mask: u8 [N] or NULL — robots whose history is first set to x (init_history; the filtered value follows from it);
 * mode 0 = filter, 1 = reset (histories zeroed for robots in init_mask, or all if NULL; x / y unused). */
int mg_a1_action_filter(const mg_a1_filter_config *cfg, int32_t n_envs, double *xhist, double *yhist, const double *x,
                        double *y, const uint8_t *init_mask, int32_t mode, void *stream);

/* The Python-computed entries of LocomotionGymEnv's `info` (locomotion_gym_env.py:534-545) from the control observation:
 *   pose          GetBaseRollPitchYaw minitaur.py:622-636 — roll, pitch, yaw of the DELAYED base quaternion. The reference
 *                 asks Bullet (getEulerFromQuaternion); here the standard ZYX formulas (roll = atan2(2(wx+yz), 1-2(x^2+y^2)),
 *                 pitch = asin(2(wy-zx)), yaw = atan2(2(wz+xy), 1-2(y^2+z^2))) — Bullet's own code is not in the reference tree
 *   rot_mat       getMatrixFromQuaternion(GetBaseOrientation()) :830-838: matrix of the quaternion rebuilt from `pose`
 *   footposition  GetFootPositionsInBaseFrame a1.py:141-147,527-530: leg forward kinematics of the motor angles (a1.py:105-123)
 *   joint_angle, drpy, energy as mg_a1_sensors.
 * Every output may be NULL. pose [3][N], rot_mat [9][N], footposition [12][N], joint_angle [12][N], drpy [3][N], energy [N]. */
int mg_a1_info(const mg_a1_actuator_config *cfg, int32_t n_envs, const mg_a1_actuator_state *state, double *pose,
               double *rot_mat, double *footposition, double *joint_angle, double *drpy, double *energy, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* METAGYM_HIP_H */
