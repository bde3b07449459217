// graph_probe.hip -- is a hipGraph of an evaluation's launch chain worth anything on this box?  (round 5)
//
// An objective evaluation is a chain of 6 dependent launches (matrices, three table levels, walk, score) that ends with a
// word in pinned host memory; the host then launches the next chain.  tools/gate_probe.hip measured 26.7 us for six
// dependent TINY launches -> word visible, i.e. ~4 us per launch boundary on the device side.  This probe asks whether the
// same six launches replayed from an instantiated hipGraph (one hipGraphLaunch per evaluation) are any cheaper:
//   S  six launches on a stream, one after the other (today)
//   G  the same six captured once, replayed with hipGraphLaunch
// each with kernels that (a) return at once, (b) last ~10 us each (256 workgroups spinning on the wall clock), so that the
// host's enqueue runs ahead of the device as it does in a real evaluation and only the device-side boundaries remain.
// Reported: host "go" -> word visible, per chain, and the same minus the kernels' own time.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/graph_probe tools/graph_probe.hip && tools/graph_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

struct Words {
    volatile unsigned long long ack;   // device -> host
    unsigned long long pad[7];
};

// a link of the chain: every workgroup spins `ticks` of the 100 MHz clock, one atomic per workgroup
__global__ void k_link(int* counter, long long ticks)
{
    if (ticks > 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    }
    if (threadIdx.x == 0) atomicAdd(counter, 1);
}

// the chain's last launch: the next sequence number (kept on the device: a graph's arguments are baked) to the host
__global__ void k_ack_next(Words* w, unsigned long long* dseq, long long ticks)
{
    if (ticks > 0) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long v = *dseq + 1;
        *dseq = v;
        __hip_atomic_store(&w->ack, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void busy_wait_us(double us)
{
    const double t0 = now_us();
    while (now_us() - t0 < us) {}
}
static void report(const char* what, std::vector<double>& v, double kernels_us)
{
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    printf("%-64s median %7.2f us   p10 %7.2f   p90 %7.2f   minus the kernels' %5.1f us: %6.2f   (n=%zu)\n", what, med, v[v.size() / 10],
           v[v.size() * 9 / 10], kernels_us, med - kernels_us, v.size());
}

static void enqueue_chain(hipStream_t st, int links, int* d_counter, Words* w, unsigned long long* d_seq, long long ticks)
{
    for (int k = 0; k < links; ++k) hipLaunchKernelGGL(k_link, dim3(256), dim3(256), 0, st, d_counter, ticks);
    hipLaunchKernelGGL(k_ack_next, dim3(1), dim3(64), 0, st, w, d_seq, ticks);
}

int main()
{
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Words* w = nullptr;
    CK(hipHostMalloc((void**)&w, sizeof(Words), hipHostMallocMapped | hipHostMallocCoherent));
    w->ack = 0;
    int* d_counter = nullptr;
    unsigned long long* d_seq = nullptr;
    CK(hipMalloc((void**)&d_counter, sizeof(int)));
    CK(hipMalloc((void**)&d_seq, sizeof(unsigned long long)));
    CK(hipMemset(d_counter, 0, sizeof(int)));
    CK(hipMemset(d_seq, 0, sizeof(unsigned long long)));
    CK(hipDeviceSynchronize());
    const int N = 400, LINKS = 5;
    unsigned long long seq = 0;

    for (long long ticks : {0LL, 1000LL}) {
        const double kernels_us = (LINKS + 1) * (double)ticks / 100.0;
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        enqueue_chain(st, LINKS, d_counter, w, d_seq, ticks);
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 50; ++i) {   // warm-up of both forms
            ++seq;
            if (i & 1) CK(hipGraphLaunch(exec, st));
            else enqueue_chain(st, LINKS, d_counter, w, d_seq, ticks);
            while (w->ack != seq) {}
        }
        std::vector<double> s, g, s_back, g_back;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < N; ++i) {
                ++seq;
                busy_wait_us(20);
                const double t0 = now_us();
                enqueue_chain(st, LINKS, d_counter, w, d_seq, ticks);
                while (w->ack != seq) {}
                s.push_back(now_us() - t0);
            }
            for (int i = 0; i < N; ++i) {
                ++seq;
                busy_wait_us(20);
                const double t0 = now_us();
                CK(hipGraphLaunch(exec, st));
                while (w->ack != seq) {}
                g.push_back(now_us() - t0);
            }
        }
        // back to back, as an optimiser drives it: no pause between the word and the next chain; period per chain
        {
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) {
                ++seq;
                enqueue_chain(st, LINKS, d_counter, w, d_seq, ticks);
                while (w->ack != seq) {}
            }
            s_back.push_back((now_us() - t0) / N);
            const double t1 = now_us();
            for (int i = 0; i < N; ++i) {
                ++seq;
                CK(hipGraphLaunch(exec, st));
                while (w->ack != seq) {}
            }
            g_back.push_back((now_us() - t1) / N);
        }
        printf("-- chain of %d launches, each kernel %s\n", LINKS + 1, ticks ? "~10 us (256 workgroups on the wall clock)" : "returns at once");
        report("S  six launches on the stream -> word visible", s, kernels_us);
        report("G  one hipGraphLaunch of the captured six -> word visible", g, kernels_us);
        printf("   back to back (no pause between chains), period per chain: stream %.2f us, graph %.2f us\n", s_back[0], g_back[0]);
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
    }
    CK(hipStreamSynchronize(st));
    return 0;
}
