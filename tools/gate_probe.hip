// gate_probe.hip -- what a host-released, pre-enqueued launch chain is worth on this box (round 5).
//
// Every objective evaluation ends with the host reading a word the score kernel wrote to pinned memory and starts with the
// host launching the next chain.  This probe times the round trip "host decides -> a kernel of the chain has run and its
// word is visible to the host" four ways:
//   A  hipLaunchKernel of one tiny kernel (today's first launch of an evaluation)
//   B  the kernel was enqueued earlier behind a one-wave GATE kernel that polls a word in pinned host memory;
//      the host releases it with one store
//   C  the same with hipStreamWaitValue64 in place of the gate kernel (if the device reports support)
//   D  A and B with a chain of 5 dependent tiny kernels behind the first (launch gaps of a whole evaluation)
// and the cost of a CANCELLED chain (gate says no: 5 kernels that exit at once).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/gate_probe tools/gate_probe.hip && tools/gate_probe
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
    volatile unsigned long long flag;   // host -> device
    unsigned long long pad0[7];
    volatile unsigned long long ack;    // device -> host
    unsigned long long pad1[7];
    volatile unsigned long long expired;
};

__global__ void k_ack(Words* w, unsigned long long v, const int* go)
{
    if (go && __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&w->ack, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void k_link(int* counter, const int* go)
{
    if (go && __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    if (threadIdx.x == 0) atomicAdd(counter, 1);
}

// one wave: wait until the host's word equals `want` (go) or `want | 1<<63` (cancel), at most `ticks` of the 100 MHz clock
__global__ void k_gate(Words* w, unsigned long long want, long long ticks, int* go)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    int decision = 0;
    for (;;) {
        const unsigned long long f = __hip_atomic_load(&w->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (f == want) { decision = 1; break; }
        if (f == (want | (1ull << 63))) break;
        if (wall_clock64() - t0 > ticks) {
            __hip_atomic_store(&w->expired, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(go, decision, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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
static void report(const char* what, std::vector<double>& v)
{
    std::sort(v.begin(), v.end());
    printf("%-78s median %7.2f us   p10 %7.2f   p90 %7.2f   (n=%zu)\n", what, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10], v.size());
}

int main()
{
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Words* w = nullptr;
    CK(hipHostMalloc((void**)&w, sizeof(Words), hipHostMallocMapped | hipHostMallocCoherent));
    w->flag = 0;
    w->ack = 0;
    w->expired = 0;
    int *d_go = nullptr, *d_counter = nullptr;
    CK(hipMalloc((void**)&d_go, sizeof(int)));
    CK(hipMalloc((void**)&d_counter, sizeof(int)));
    CK(hipMemset(d_go, 0, sizeof(int)));
    CK(hipMemset(d_counter, 0, sizeof(int)));
    CK(hipDeviceSynchronize());
    const int N = 400;
    unsigned long long seq = 0;
    const int* no_go = nullptr;

    // warm-up
    for (int i = 0; i < 50; ++i) {
        ++seq;
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, no_go);
        while (w->ack != seq) {}
    }
    std::vector<double> a, b, c, d0, d1, e;
    // A: plain launch
    for (int i = 0; i < N; ++i) {
        ++seq;
        busy_wait_us(30);
        const double t0 = now_us();
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, no_go);
        while (w->ack != seq) {}
        a.push_back(now_us() - t0);
    }
    report("A  hipLaunchKernel(tiny) -> word visible on the host", a);
    // B: pre-enqueued behind a gate kernel
    for (int i = 0; i < N; ++i) {
        ++seq;
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, w, seq, (long long)2e5, d_go);
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, (const int*)d_go);
        busy_wait_us(30);   // the gate is spinning by now
        const double t0 = now_us();
        w->flag = seq;
        while (w->ack != seq) {}
        b.push_back(now_us() - t0);
    }
    report("B  store to a pinned word -> gate kernel -> pre-enqueued tiny kernel -> word visible", b);
    // C: hipStreamWaitValue64
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (can) {
        unsigned long long* sig = nullptr;
        hipError_t er = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
        printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(er));
        bool ok = true;
        for (int pass = 0; pass < 2 && ok; ++pass) {
            // pass 0: wait on the pinned host word; pass 1: wait on signal memory (host writes through the pointer)
            volatile unsigned long long* target = pass == 0 ? &w->flag : (volatile unsigned long long*)sig;
            if (pass == 1 && er != hipSuccess) break;
            c.clear();
            for (int i = 0; i < N; ++i) {
                ++seq;
                hipError_t e1 = hipStreamWaitValue64(st, (void*)target, seq, hipStreamWaitValueEq, ~0ull);
                if (e1 != hipSuccess) {
                    printf("hipStreamWaitValue64 (%s): %s\n", pass ? "signal memory" : "pinned host word", hipGetErrorString(e1));
                    ok = pass == 0;   // try the other target
                    break;
                }
                hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, no_go);
                busy_wait_us(30);
                const double t0 = now_us();
                *target = seq;
                const double tmax = t0 + 2e6;
                while (w->ack != seq && now_us() < tmax) {}
                if (w->ack != seq) {
                    printf("hipStreamWaitValue64 (%s): never released\n", pass ? "signal memory" : "pinned host word");
                    return 2;
                }
                c.push_back(now_us() - t0);
            }
            if (!c.empty())
                report(pass ? "C2 store to SIGNAL memory -> hipStreamWaitValue64 -> tiny kernel -> word visible" : "C1 store to a pinned word -> hipStreamWaitValue64 -> tiny kernel -> word visible", c);
        }
    }
    // D: a chain of 6 dependent kernels (5 links + ack): launched now  vs  released by the gate
    for (int i = 0; i < N; ++i) {
        ++seq;
        busy_wait_us(30);
        const double t0 = now_us();
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_link, dim3(256), dim3(256), 0, st, d_counter, no_go);
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, no_go);
        while (w->ack != seq) {}
        d0.push_back(now_us() - t0);
    }
    report("D0 6 dependent launches now -> word visible", d0);
    for (int i = 0; i < N; ++i) {
        ++seq;
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, w, seq, (long long)2e5, d_go);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_link, dim3(256), dim3(256), 0, st, d_counter, (const int*)d_go);
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, (const int*)d_go);
        busy_wait_us(40);
        const double t0 = now_us();
        w->flag = seq;
        while (w->ack != seq) {}
        d1.push_back(now_us() - t0);
    }
    report("D1 the same 6 launches pre-enqueued behind the gate, released by one store", d1);
    // E: a cancelled chain: gate says no, 6 kernels exit at once; time until the stream is idle again
    for (int i = 0; i < N; ++i) {
        ++seq;
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, w, seq, (long long)2e5, d_go);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_link, dim3(256), dim3(256), 0, st, d_counter, (const int*)d_go);
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, (const int*)d_go);
        busy_wait_us(40);
        const double t0 = now_us();
        w->flag = seq | (1ull << 63);
        ++seq;
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, no_go);   // what the host would launch instead
        while (w->ack != seq) {}
        e.push_back(now_us() - t0);
    }
    report("E  cancel a pre-enqueued chain of 6 + a plain launch behind it -> word visible", e);
    // F: gate expiry: nobody releases; the gate gives up after 200 us and says so
    {
        ++seq;
        w->expired = 0;
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, st, w, seq, (long long)2e4, d_go);
        hipLaunchKernelGGL(k_ack, dim3(1), dim3(64), 0, st, w, seq, (const int*)d_go);
        const double t0 = now_us();
        while (w->expired != seq && now_us() - t0 < 1e6) {}
        printf("F  unreleased gate with a 200 us slice: expired word seen after %.1f us, ack %s\n", now_us() - t0, w->ack == seq ? "WRITTEN (wrong)" : "not written (right)");
    }
    CK(hipStreamSynchronize(st));
    return 0;
}
