// Bring-up harness of the tile-dataflow Cholesky: one flow_solve of a small SPD system under a host-side watchdog (a launch that does
// not finish within 5 s is reported and the process leaves without waiting for it), residual check against the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bundler_sfm_amd/csrc scripts/r4/flow_dbg.hip -o scripts/r4/_build/flow_dbg
#include "chol_flow.hip.h"
#include <chrono>
#include <thread>
#include <unistd.h>
namespace bsfm {
hipError_t dev_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes); }
void dev_free(void* p, bool) { if (p) (void)hipFree(p); }
}
using namespace bsfm;

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int ld = (n + 127) / 128 * 128;
    std::vector<double> A((size_t)ld * ld, 0.0), b(ld, 0.0), x(n);
    unsigned long long sd = 88172645463325252ull;
    auto rnd = [&] { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (double)(sd >> 11) / 9007199254740992.0 - 0.5; };
    for (int r = 0; r < n; ++r) for (int c = 0; c <= r; ++c) { const double v = rnd(); A[(size_t)r * ld + c] = v; A[(size_t)c * ld + r] = v; }
    for (int r = 0; r < n; ++r) { A[(size_t)r * ld + r] = 0.6 * n + 1.0 + r * 0.01; b[r] = rnd(); }
    PotrfWorkspace ws;
    if (potrf_init(ws, ld, 0)) { printf("potrf_init failed\n"); return 1; }
    ws.timing = 0;
    double *dS, *dE, *dx; int* dinfo;
    hipMalloc(&dS, (size_t)ld * ld * 8); hipMalloc(&dE, ld * 8); hipMalloc(&dx, ld * 8); hipMalloc(&dinfo, 4);
    hipMemset(dinfo, 0, 4);
    hipMemcpy(dS, A.data(), (size_t)ld * ld * 8, hipMemcpyHostToDevice);
    hipMemcpy(dE, b.data(), ld * 8, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    ws.flow = new FlowWorkspace();
    hipDeviceSynchronize();
    const int rc = flow_solve(ws, *ws.flow, dS, ld, n, dE, dx, dinfo, st);
    printf("n = %d: %zu tasks (%zu chain), flow_solve %d, launch %s\n", n, ws.flow->sched.tasks.size(), ws.flow->chain.size(), rc, hipGetErrorString(hipGetLastError()));
    bool done = false; hipError_t qe = hipSuccess;
    for (int it = 0; it < 500 && !done; ++it) { std::this_thread::sleep_for(std::chrono::milliseconds(10)); qe = hipStreamQuery(st); done = qe == hipSuccess; }
    if (!done) { printf("STUCK (hipStreamQuery: %s)\n", hipGetErrorString(qe)); fflush(stdout); _exit(2); }
    int info = 0; hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost); hipMemcpy(x.data(), dx, n * 8, hipMemcpyDeviceToHost);
    double rmax = 0, xmax = 0;
    for (int r = 0; r < n; ++r) { double s = -b[r]; for (int c = 0; c < n; ++c) s += A[(size_t)r * ld + c] * x[c]; rmax = std::max(rmax, fabs(s)); xmax = std::max(xmax, fabs(x[r])); }
    printf("info %d, residual max %.3e, |x| max %.3e %s\n", info, rmax, xmax, rmax < 1e-10 ? "OK" : "BAD");
    return 0;
}
