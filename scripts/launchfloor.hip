// Diagnostic: what a dependent kernel launch costs on one stream - plain launches against the
// same chain replayed from a captured graph, for an empty kernel and for one that touches memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__global__ void k_empty(uint32_t *p)
{
    if (p == nullptr)
        return;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        p[0] += 1u;
}

__global__ void k_touch(uint32_t *p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] += 1u;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const uint32_t n = 1u << 20;
    uint32_t *d;
    CHECK(hipMalloc(&d, 4 * n));
    CHECK(hipMemset(d, 0, 4 * n));
    const int chain = argc > 1 ? atoi(argv[1]) : 200; /* launches per chain */
    for (int variant = 0; variant < 3; ++variant) {
        auto enqueue = [&]() {
            for (int i = 0; i < chain; ++i) {
                if (variant == 0)
                    k_empty<<<1, 64, 0, s>>>(d);
                else if (variant == 1)
                    k_touch<<<256, 256, 0, s>>>(d, 65536);
                else
                    k_touch<<<n / 256, 256, 0, s>>>(d, n);
            }
        };
        enqueue();
        CHECK(hipStreamSynchronize(s));
        double best = 1e30;
        for (int r = 0; r < 50; ++r) {
            const double t0 = now_us();
            enqueue();
            CHECK(hipStreamSynchronize(s));
            const double t = now_us() - t0;
            if (t < best)
                best = t;
        }
        hipGraph_t g;
        hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue();
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipStreamSynchronize(s));
        double bestg = 1e30;
        for (int r = 0; r < 50; ++r) {
            const double t0 = now_us();
            CHECK(hipGraphLaunch(ge, s));
            CHECK(hipStreamSynchronize(s));
            const double t = now_us() - t0;
            if (t < bestg)
                bestg = t;
        }
        printf("variant %d (%s), chains of %d: plain %.2f us per launch (%.1f us a chain), graph %.2f us per launch (%.1f us a chain)\n",
               variant, variant == 0 ? "one wave" : variant == 1 ? "256 blocks, 256 KB" : "4096 blocks, 4 MB", chain,
               best / chain, best, bestg / chain, bestg);
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
