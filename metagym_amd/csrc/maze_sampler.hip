// maze_sampler.hip — on-device MetaMaze task generation: new tasks are drawn
// straight into the device task table, no host round trip.
//
// Replaces MazeTaskManager.sample_task (reference metagym/metamaze/envs/maze_task.py:41-190) for a
// whole table of tasks. Task t is, bit for bit, what the reference returns after
//     random.seed(seed_t); numpy.random.seed(seed_t); sample_task(**params)
// which requires the two MT19937 streams the reference consumes (python `random`: init_by_array,
// _randbelow / shuffle / random; numpy legacy RandomState: init_genrand, masked-rejection randint,
// rand) and numpy's pairwise float64 summation for the food loop's termination test. The CPU
// restatement with the citations is oracle/maze_sampler.py; tests compare both with tasks drawn by
// the unmodified reference (tests/golden/maze_tasks.npz).
//
// Mapping: the algorithm is a serial program per task (Fisher-Yates shuffles and a Prim-style
// wall digger driven by one random stream), so parallelism is ACROSS tasks: one wave per task, all
// working state in LDS (two 624-word generator states, wall / component grids, the wall list, the
// food array). Every lane of the wave runs the same uniform program — values are wave-uniform, LDS
// writes of a uniform value to a uniform address are benign — and the lane id is used only where
// the work is data-parallel (grid initialisation, component relabelling, writing the table rows).
#include "mg_common.h"

namespace {

constexpr int MTN = 624, MTM = 397;

struct MT {
    uint32_t *mt;   // LDS [624]
    int idx;
};

__device__ __forceinline__ void mt_init_genrand(MT &g, uint32_t s) {
    g.mt[0] = s;
    for (int i = 1; i < MTN; ++i) {
        const uint32_t p = g.mt[i - 1];
        g.mt[i] = 1812433253u * (p ^ (p >> 30)) + (uint32_t)i;
    }
    g.idx = MTN;
}

// CPython random.seed(int): key = 32-bit little-endian words of |seed| (one word for seed < 2^32)
__device__ __forceinline__ void mt_init_by_array1(MT &g, uint32_t key0) {
    mt_init_genrand(g, 19650218u);
    int i = 1;
    for (int k = MTN; k > 0; --k) {                     // key length 1: key[j] + j == key0 throughout
        const uint32_t p = g.mt[i - 1];
        g.mt[i] = (g.mt[i] ^ ((p ^ (p >> 30)) * 1664525u)) + key0;
        ++i;
        if (i >= MTN) { g.mt[0] = g.mt[MTN - 1]; i = 1; }
    }
    for (int k = MTN - 1; k > 0; --k) {
        const uint32_t p = g.mt[i - 1];
        g.mt[i] = (g.mt[i] ^ ((p ^ (p >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= MTN) { g.mt[0] = g.mt[MTN - 1]; i = 1; }
    }
    g.mt[0] = 0x80000000u;
    g.idx = MTN;
}

__device__ __forceinline__ uint32_t mt_next(MT &g) {
    if (g.idx >= MTN) {
        for (int k = 0; k < MTN; ++k) {
            const uint32_t y = (g.mt[k] & 0x80000000u) | (g.mt[k + 1 == MTN ? 0 : k + 1] & 0x7fffffffu);
            g.mt[k] = g.mt[k + MTM >= MTN ? k + MTM - MTN : k + MTM] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g.idx = 0;
    }
    uint32_t y = g.mt[g.idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// genrand_res53: python random.random() and numpy's legacy double
__device__ __forceinline__ double mt_double(MT &g) {
    const uint32_t a = mt_next(g) >> 5, b = mt_next(g) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

// Lib/random.py _randbelow_with_getrandbits
__device__ __forceinline__ int py_randbelow(MT &g, int n) {
    const int k = 32 - __clz(n);
    uint32_t r = mt_next(g) >> (32 - k);
    while (r >= (uint32_t)n) r = mt_next(g) >> (32 - k);
    return (int)r;
}

// numpy's float64 add-reduce over a contiguous run (loops_utils.h pairwise_sum): runs of <= 128
// elements use 8 accumulators; longer runs are halved (first half rounded down to a multiple of 8).
__device__ double pairwise_block(const double *a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 = r0 + a[i]; r1 = r1 + a[i + 1]; r2 = r2 + a[i + 2]; r3 = r3 + a[i + 3];
        r4 = r4 + a[i + 4]; r5 = r5 + a[i + 5]; r6 = r6 + a[i + 6]; r7 = r7 + a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + a[i];
    return res;
}

__device__ double pairwise_sum(const double *a, int n) {
    // post-order walk of  sum(lo, n) = sum(lo, n2) + sum(lo + n2, n - n2)  with an explicit stack
    // (n <= 63 * 63 splits at most 5 times)
    int lo[8], len[8], state[8];
    double left[8];
    int sp = 0;
    lo[0] = 0; len[0] = n; state[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        if (len[sp] <= 128) {
            ret = pairwise_block(a + lo[sp], len[sp]);
            --sp;
            continue;
        }
        int n2 = len[sp] / 2;
        n2 -= n2 % 8;
        if (state[sp] == 0) {
            state[sp] = 1;
            lo[sp + 1] = lo[sp]; len[sp + 1] = n2; state[sp + 1] = 0;
            ++sp;
        } else if (state[sp] == 1) {
            left[sp] = ret;
            state[sp] = 2;
            lo[sp + 1] = lo[sp] + n2; len[sp + 1] = len[sp] - n2; state[sp + 1] = 0;
            ++sp;
        } else {
            ret = left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

struct SampleK {
    int n, allow_loops, n_texts, food_interval;
    double cell_size, wall_height, agent_height, step_reward, goal_reward, food_reward, initial_life, max_life,
        food_density, crowd_ratio;
    int has_goal_reward;
    uint32_t seed_base;
};

__global__ __launch_bounds__(64) void maze_sample_tasks_kernel(SampleK k, int n_tasks, const uint32_t *seeds,
                                                              int32_t *start, int32_t *goal, int8_t *walls_out,
                                                              uint8_t *texts_out, double *food_out,
                                                              int32_t *interval_out, double *scalars) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= n_tasks) return;
    const int n = k.n, nn = n * n;
    double *food = reinterpret_cast<double *>(smem);
    uint32_t *mt_py = reinterpret_cast<uint32_t *>(food + nn);
    uint32_t *mt_np = mt_py + MTN;
    int16_t *path = reinterpret_cast<int16_t *>(mt_np + MTN);      // component id per cell (interior free cells)
    int16_t *wall_list = path + nn;                                  // interior wall cells, dict order
    int16_t *order = wall_list + nn;                                 // shuffled copy
    int8_t *walls = reinterpret_cast<int8_t *>(order + nn);
    uint8_t *texts = reinterpret_cast<uint8_t *>(walls + nn);

    const uint32_t seed = seeds != nullptr ? seeds[t] : k.seed_base + (uint32_t)t;
    MT py{mt_py, 0}, np_{mt_np, 0};
    mt_init_by_array1(py, seed);
    mt_init_genrand(np_, seed);

    // cell_texts = numpy.random.randint(1, n_texts, (n, n))  (:61): masked rejection on 32-bit draws
    {
        const uint32_t rng = (uint32_t)(k.n_texts - 2);
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        for (int c = 0; c < nn; ++c) {
            uint32_t v = 0;
            if (rng != 0) {
                v = mt_next(np_) & mask;
                while (v > rng) v = mt_next(np_) & mask;
            }
            texts[c] = (uint8_t)(1u + v);
        }
    }
    for (int c = lane; c < nn; c += 64) {                                          // :60,:64-66
        const int i = c / n, j = c - i * n;
        walls[c] = ((i & 1) && (j & 1)) ? 0 : 1;
    }
    __syncthreads();
    const int m = (n - 1) / 2;
    const int s_x = py_randbelow(py, m) * 2 + 1;                                   // :68-69
    const int s_y = py_randbelow(py, m) * 2 + 1;
    int g_x = n - 2, g_y = n - 2;
    const double min_dist = 0.45 * (double)n;
    for (int a = 0; a < m; ++a)                                                    // :75-83: `break` leaves the inner loop only
        for (int b = 0; b < m; ++b) {
            const int e_x = py_randbelow(py, m) * 2 + 1;
            const int e_y = py_randbelow(py, m) * 2 + 1;
            const int d2 = (e_x - s_x) * (e_x - s_x) + (e_y - s_y) * (e_y - s_y);
            if (sqrt((double)d2) > min_dist) { g_x = e_x; g_y = e_y; break; }
        }
    // :86-98 wall dict (row-major insertion order) and one component per free interior cell
    int n_walls = 0, n_paths = 0;
    for (int i = 1; i < n - 1; ++i)
        for (int j = 1; j < n - 1; ++j) {
            const int c = i * n + j;
            if (walls[c] > 0) wall_list[n_walls++] = (int16_t)c;
            else path[c] = (int16_t)n_paths++;
        }
    const double wall_budget = (double)((n - 2) * (n - 2)) * k.crowd_ratio;
    // :103 Prim-style digging until one component remains (and, with loops, the wall share is low enough)
    while (n_paths > 1 || (k.allow_loops && (double)n_walls > wall_budget)) {
        for (int q = lane; q < n_walls; q += 64) order[q] = wall_list[q];
        __syncthreads();
        for (int q = n_walls - 1; q > 0; --q) {                                    // random.shuffle
            const int r = py_randbelow(py, q + 1);
            const int16_t tmp = order[q];
            order[q] = order[r];
            order[r] = tmp;
        }
        int new_id = -1, cell = -1, ab0 = -1, ab1 = -1, ab2 = -1, n_ab = 0;
        for (int q = 0; q < n_walls; ++q) {
            cell = order[q];
            const int i = cell / n, j = cell - i * n;
            new_id = -1; ab0 = ab1 = ab2 = -1; n_ab = 0;
            int ids[4], n_ids = 0, max_dup = 1;
            const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
            for (int d = 0; d < 4; ++d) {
                const int a = i + di[d], b = j + dj[d];
                if (a > 0 && a < n && b > 0 && b < n && walls[a * n + b] < 1) {
                    const int pid = path[a * n + b];
                    int dup = 1;
                    for (int u = 0; u < n_ids; ++u) dup += ids[u] == pid;
                    ids[n_ids++] = pid;
                    max_dup = dup > max_dup ? dup : max_dup;
                    int drop = -1;
                    if (pid < new_id || new_id < 0) {
                        drop = new_id;
                        new_id = pid;
                    } else if (pid != new_id) {
                        drop = pid;
                    }
                    if (drop >= 0 && drop != ab0 && drop != ab1 && drop != ab2) {
                        if (n_ab == 0) ab0 = drop; else if (n_ab == 1) ab1 = drop; else ab2 = drop;
                        ++n_ab;
                    }
                }
            }
            if (n_ab >= 1 && max_dup < 2) break;
            if (n_ab >= 1 && max_dup > 1 && k.allow_loops) break;
            if (k.allow_loops && n_paths < 2 && mt_double(py) < 0.2) break;
        }
        if (new_id < 0) continue;
        // :144-154 release the wall, merge the abandoned components into new_id
        path[cell] = (int16_t)new_id;
        walls[cell] = 0;
        {   // del wall_dict[i, j] keeps the order of the rest
            int pos = 0;
            while (wall_list[pos] != cell) ++pos;
            __syncthreads();
            for (int q = pos + lane; q < n_walls - 1; q += 64) order[q] = wall_list[q + 1];
            __syncthreads();
            for (int q = pos + lane; q < n_walls - 1; q += 64) wall_list[q] = order[q];
            --n_walls;
        }
        if (n_ab > 0) {
            for (int c = lane; c < nn; c += 64) {
                const int p = path[c];
                if (walls[c] < 1 && (p == ab0 || p == ab1 || p == ab2)) path[c] = (int16_t)new_id;
            }
            n_paths -= n_ab;
        }
        __syncthreads();
    }
    // :157-160 corridors get the ground texture (interior only)
    for (int c = lane; c < nn; c += 64) {
        const int i = c / n, j = c - i * n;
        if (i > 0 && i < n - 1 && j > 0 && j < n - 1 && walls[c] < 1) texts[c] = 0;
    }
    // :164-169 food: clip(rand * food_reward, 0.10, food_reward) * (1 - wall), thinned by 10 % per
    // round until numpy.sum(food) <= (n-1)^2 * food_density
    for (int c = 0; c < nn; ++c) {
        double f = mt_double(np_) * k.food_reward;
        f = f < 0.10 ? 0.10 : f;                  // numpy.clip = minimum(maximum(x, lo), hi)
        f = f > k.food_reward ? k.food_reward : f;
        food[c] = f * (1.0 - (double)walls[c]);
    }
    const double exp_food = (double)((n - 1) * (n - 1)) * k.food_density;
    while (0.0 + pairwise_sum(food, nn) > exp_food)
        for (int c = 0; c < nn; ++c) food[c] = food[c] * (mt_double(np_) < 0.90 ? 1.0 : 0.0);
    __syncthreads();

    // ---- table rows ---------------------------------------------------------------------------
    const size_t row = (size_t)t * nn;
    for (int c = lane; c < nn; c += 64) {
        walls_out[row + c] = walls[c];
        texts_out[row + c] = texts[c];
        food_out[row + c] = food[c];
        interval_out[row + c] = food[c] > 1.0e-3 ? k.food_interval : 0;
    }
    if (lane == 0) {
        start[2 * t] = s_x; start[2 * t + 1] = s_y;
        goal[2 * t] = g_x; goal[2 * t + 1] = g_y;
        double *s = scalars + 8 * (size_t)t;
        s[0] = k.cell_size; s[1] = k.wall_height; s[2] = k.agent_height; s[3] = k.initial_life; s[4] = k.max_life;
        s[5] = k.step_reward;
        // :163  - numpy.sqrt(n) * n * step_reward, left to right
        s[6] = k.has_goal_reward ? k.goal_reward : ((-sqrt((double)n)) * (double)n) * k.step_reward;
        s[7] = 0.0;
    }
}

}  // namespace

extern "C" int mg_maze_sample_tasks(const mg_maze_sample_params *p, int32_t n_tasks, uint32_t seed_base,
                                    const uint32_t *seeds, int32_t *start, int32_t *goal, int8_t *walls,
                                    uint8_t *texts, double *food_rewards, int32_t *food_interval, double *scalars,
                                    void *stream) {
    MG_REQUIRE_PTR(p);
    MG_REQUIRE_PTR(start); MG_REQUIRE_PTR(goal); MG_REQUIRE_PTR(walls); MG_REQUIRE_PTR(texts);
    MG_REQUIRE_PTR(food_rewards); MG_REQUIRE_PTR(food_interval); MG_REQUIRE_PTR(scalars);
    if (n_tasks <= 0) return mg::set_error(MG_ERR_BAD_SIZE, "mg_maze_sample_tasks: n_tasks = %d", n_tasks);
    // the reference's own asserts (maze_task.py:56-57,161-162) plus the table limits of this kernel
    if (p->n <= 6) return mg::set_error(MG_ERR_BAD_CONFIG, "Minimum required cells are 7 (got n = %d)", p->n);
    if (p->n % 2 == 0) return mg::set_error(MG_ERR_BAD_CONFIG, "Cell Numbers can only be odd (got n = %d)", p->n);
    if (p->n > 63) return mg::set_error(MG_ERR_UNSUPPORTED, "mg_maze_sample_tasks: n = %d > 63", p->n);
    if (!(p->step_reward < 0)) return mg::set_error(MG_ERR_BAD_CONFIG, "step_reward must be < 0");
    if (p->has_goal_reward && !(p->goal_reward > 0)) return mg::set_error(MG_ERR_BAD_CONFIG, "goal reward must be > 0");
    if (p->n_texts < 2 || p->n_texts > 255)
        return mg::set_error(MG_ERR_BAD_CONFIG, "mg_maze_sample_tasks: n_texts = %d (need 2..255)", p->n_texts);
    if (!(p->food_reward >= 0.10))   // numpy.clip with lo > hi returns hi everywhere; not a configuration the reference uses
        return mg::set_error(MG_ERR_UNSUPPORTED, "mg_maze_sample_tasks: food_reward < 0.10");
    if (!(p->food_density >= 0.0) || !(p->crowd_ratio >= 0.0))
        return mg::set_error(MG_ERR_BAD_CONFIG, "mg_maze_sample_tasks: food_density and crowd_ratio must be >= 0");
    SampleK k{};
    k.n = p->n; k.allow_loops = p->allow_loops != 0; k.n_texts = p->n_texts; k.food_interval = p->food_interval;
    k.cell_size = p->cell_size; k.wall_height = p->wall_height; k.agent_height = p->agent_height;
    k.step_reward = p->step_reward; k.goal_reward = p->goal_reward; k.food_reward = p->food_reward;
    k.initial_life = p->initial_life; k.max_life = p->max_life; k.food_density = p->food_density;
    k.crowd_ratio = p->crowd_ratio; k.has_goal_reward = p->has_goal_reward != 0; k.seed_base = seed_base;
    const int nn = p->n * p->n;
    const size_t lds = sizeof(double) * nn + sizeof(uint32_t) * 2 * MTN + sizeof(int16_t) * 3 * nn + 2 * (size_t)nn;
    mg::DeviceGuard guard(mg::device_of(start));
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(maze_sample_tasks_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return mg::check_hip(e, "hipFuncSetAttribute(maze_sample_tasks_kernel)");
    }
    hipLaunchKernelGGL(maze_sample_tasks_kernel, dim3(n_tasks), dim3(64), lds, static_cast<hipStream_t>(stream), k,
                       n_tasks, seeds, start, goal, walls, texts, food_rewards, food_interval, scalars);
    return mg::check_launch("maze_sample_tasks_kernel");
}
