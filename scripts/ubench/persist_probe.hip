// Micro-benchmark (not product code): the signalling floor of a PERSISTENT step kernel on one MI355X.
//
// A resident kernel (grid x 256 threads, default 256 blocks = one wave per SIMD like the 65 536-env quadrotor launch) waits per
// "step" on a command word, does nothing (or a busy loop of `work` iterations), and acknowledges by bumping an arrival counter;
// the last block of a step publishes the step number in an ack word. The host drives steps in STREAM ORDER on a second stream:
//     hipStreamWriteValue32(S, cmd, k)  ->  [persistent kernel sees k]  ->  hipStreamWaitValue32(S, ack, k, >=)  ->  next op on S
// and times K such steps with HIP events on S. That per-step figure is what a persistent env.step() would pay on top of its
// arithmetic, to be compared with the ~5.5 us per step a launch costs the quadrotor kernel (profiles/r05/quad_rollout.txt).
//
//   persist_probe [steps=2000] [blocks=256] [work=0] [mode] [tree=0] [coarse=0]
//   mode 0: stream write + stream wait (ack in hipMallocSignalMemory)      mode 1: stream write + tiny consumer kernel that spins on ack
//   mode 2: host writes cmd in pinned host memory, host spins on ack in pinned host memory (no stream ops at all)
//   mode 3: a one-thread kernel posts cmd, a one-thread kernel spins on ack (two launches per step, no stream memory operations)
// Every wait in the kernel has a wall-clock watchdog (2 s): a protocol error ends the kernel instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %d (%s)\n", #x, (int)e_, hipGetErrorString(e_)); return 1; } } while (0)

struct Ctrl {
    unsigned cmd;          // last step requested (written in stream order)
    unsigned stop;         // 1: leave
    unsigned arrived;      // blocks that finished, all steps summed (tree: XCD groups that finished)
    unsigned timed_out;    // watchdog fired
    unsigned long long poll_cycles, polls;   // block 0: wall-clock ticks (100 MHz) spent polling, number of polls
    unsigned pad[24];
    unsigned sub[8][32];   // tree = 1: one arrival counter per blockIdx & 7 (= per XCD), each on its own 128-byte line
};

__device__ __forceinline__ unsigned ld_sys(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(256) void persistent(Ctrl *c, unsigned *ack, float *sink, int work, unsigned max_steps, int tree) {
    const unsigned nblocks = gridDim.x;
    __shared__ unsigned go;
    float acc = threadIdx.x;
    unsigned long long waited = 0, polls = 0;
    for (unsigned step = 1; step <= max_steps; ++step) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            unsigned ok = 0;
            for (;;) {
                if (ld_sys(&c->cmd) >= step) { ok = 1; break; }
                if (ld_sys(&c->stop)) break;
                ++polls;
                if (wall_clock64() - t0 > 200000000ull) {             // 2 s at 100 MHz: give up, and release every stream wait
                    atomicExch(&c->timed_out, 1u);
                    __hip_atomic_store(ack, 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            waited += wall_clock64() - t0;
            go = ok;
        }
        __syncthreads();
        if (!go) break;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                 // the step's inputs (written before cmd) are visible
        for (int i = 0; i < work; ++i) acc = acc * 1.0000001f + 0.5f;
        if (work && acc == 12345.0f) sink[threadIdx.x] = acc;
        __threadfence_system();                                   // the step's outputs before the arrival
        __syncthreads();
        if (threadIdx.x == 0) {
            if (tree && (nblocks & 7) == 0) {       // two levels: 8 groups of nblocks / 8 blocks, then the 8 group leaders
                const unsigned o1 = atomicAdd(&c->sub[blockIdx.x & 7][0], 1u);
                if (o1 + 1 == (nblocks >> 3) * step) {
                    const unsigned o2 = atomicAdd(&c->arrived, 1u);
                    if (o2 + 1 == 8u * step) __hip_atomic_store(ack, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            } else {
                const unsigned old = atomicAdd(&c->arrived, 1u);
                if (old + 1 == nblocks * step) __hip_atomic_store(ack, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { c->poll_cycles = waited; c->polls = polls; }
}

__global__ void spin_until(const unsigned *ack, unsigned want, unsigned *timed_out) {
    const unsigned long long t0 = wall_clock64();
    while (ld_sys(ack) < want) {
        if (wall_clock64() - t0 > 200000000ull) { atomicExch(timed_out, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ void post(unsigned *cmd, unsigned k) { __hip_atomic_store(cmd, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ void nop() {}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000, blocks = argc > 2 ? atoi(argv[2]) : 256, work = argc > 3 ? atoi(argv[3]) : 0;
    const int mode = argc > 4 ? atoi(argv[4]) : 0, tree = argc > 5 ? atoi(argv[5]) : 0, coarse = argc > 6 ? atoi(argv[6]) : 0;
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d; mode %d, %d steps, %d blocks x 256, work %d\n", can, mode, steps, blocks, work);
    hipStream_t K, S;
    CK(hipStreamCreateWithFlags(&K, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    Ctrl *c = nullptr;
    unsigned *ack = nullptr;
    float *sink = nullptr;
    CK(hipMalloc(&sink, 256 * sizeof(float)));
    if (mode == 2) {
        CK(hipHostMalloc((void **)&c, sizeof(Ctrl), hipHostMallocMapped | hipHostMallocCoherent));
        CK(hipHostMalloc((void **)&ack, 8, hipHostMallocMapped | hipHostMallocCoherent));
        *c = Ctrl{};
        *ack = 0;
    } else {
        hipError_t e = coarse ? hipMalloc((void **)&c, sizeof(Ctrl)) : hipExtMallocWithFlags((void **)&c, sizeof(Ctrl), hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("hipExtMallocWithFlags(uncached) -> %d (%s); falling back to hipMalloc\n", (int)e, hipGetErrorString(e)); CK(hipMalloc((void **)&c, sizeof(Ctrl))); }
        CK(hipMemset(c, 0, sizeof(Ctrl)));
        e = hipExtMallocWithFlags((void **)&ack, 8, hipMallocSignalMemory);
        if (e != hipSuccess) {
            printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %d (%s)\n", (int)e, hipGetErrorString(e));
            if (mode == 0) return 2;
            CK(hipMalloc((void **)&ack, 8));
        }
        CK(hipMemset(ack, 0, 8));
    }
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(persistent, dim3(blocks), dim3(256), 0, K, c, ack, sink, work, (unsigned)(steps + 64), tree);
    printf("arrival %s, control block in %s memory\n", tree ? "tree (8 groups by blockIdx & 7)" : "flat", coarse ? "coarse-grained (hipMalloc)" : "uncached (fine-grained)");
    CK(hipGetLastError());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned *to = &c->timed_out;
    auto run = [&](int first, int count) -> int {
        for (int k = first; k < first + count; ++k) {
            if (mode == 2) {
                __atomic_store_n(&c->cmd, (unsigned)k, __ATOMIC_RELEASE);
                const auto t0 = std::chrono::steady_clock::now();
                while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) < (unsigned)k)
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) { printf("host: ack timeout at step %d\n", k); return 1; }
                continue;
            }
            if (mode == 3) {          // no stream memory operations at all: a one-thread kernel posts the command, another one waits
                hipLaunchKernelGGL(post, dim3(1), dim3(1), 0, S, &c->cmd, (unsigned)k);
                hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, S, ack, (unsigned)k, to);
                continue;
            }
            hipError_t e = hipStreamWriteValue32(S, &c->cmd, (unsigned)k, 0);
            if (e != hipSuccess) { printf("hipStreamWriteValue32 -> %d (%s)\n", (int)e, hipGetErrorString(e)); return 1; }
            if (mode == 0) {
                e = hipStreamWaitValue32(S, ack, (unsigned)k, hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (e != hipSuccess) { printf("hipStreamWaitValue32 -> %d (%s)\n", (int)e, hipGetErrorString(e)); return 1; }
            } else {
                hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, S, ack, (unsigned)k, to);
            }
        }
        return 0;
    };
    int rc = run(1, 32);                                          // warm-up
    if (mode != 2) CK(hipStreamSynchronize(S));
    const auto h0 = std::chrono::steady_clock::now();
    if (mode != 2) CK(hipEventRecord(e0, S));
    if (!rc) rc = run(33, steps);
    if (mode != 2) { CK(hipEventRecord(e1, S)); CK(hipStreamSynchronize(S)); }
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    float ms = 0.f;
    if (mode != 2) CK(hipEventElapsedTime(&ms, e0, e1));
    // stop the resident kernel
    if (mode == 2) __atomic_store_n(&c->stop, 1u, __ATOMIC_RELEASE);
    else CK(hipStreamWriteValue32(S, &c->stop, 1u, 0));
    CK(hipDeviceSynchronize());
    Ctrl h;
    if (mode == 2) h = *c; else CK(hipMemcpy(&h, c, sizeof(Ctrl), hipMemcpyDeviceToHost));
    printf("rc %d  per step: %.3f us (HIP events on the driving stream), %.3f us (host wall)  | kernel saw cmd %u, arrived %u (= %u blocks x %u steps), "
           "watchdog %u, block 0 polled %.3f us per step in %.1f polls\n", rc, ms * 1e3 / steps, host_us / steps, h.cmd, h.arrived, blocks,
           h.arrived / (unsigned)(tree ? 8 : blocks), h.timed_out, h.poll_cycles * 0.01 / (steps + 32), (double)h.polls / (steps + 32));
    // for scale: K empty-kernel launches on one stream
    for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(nop, dim3(256), dim3(256), 0, S);
    CK(hipStreamSynchronize(S));
    CK(hipEventRecord(e0, S));
    for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(nop, dim3(256), dim3(256), 0, S);
    CK(hipEventRecord(e1, S));
    CK(hipStreamSynchronize(S));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("for scale: empty 256 x 256 kernel, back to back on one stream: %.3f us per launch\n", ms * 1e3 / steps);
    return rc;
}
