// Cost of a device-wide barrier between dependent phases: a cooperative launch of G workgroups runs K
// phases; in each phase every thread reads what its neighbour workgroup wrote in the phase before
// (through global memory), adds one and writes its own slot, then the grid synchronises.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/grid_sync.hip -o tools/micro/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void k(double *a, double *b, int phases)
{
    cg::grid_group grid = cg::this_grid();
    const int n = gridDim.x * blockDim.x;
    const int me = blockIdx.x * blockDim.x + threadIdx.x;
    const int other = ((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x;
    double *src = a, *dst = b;
    for (int p = 0; p < phases; p++) {
        dst[me] = src[other] + 1.0;
        grid.sync();
        double *t = src; src = dst; dst = t;
    }
    (void)n;
}

int main()
{
    const int phases = 200;
    for (int G : {8, 32, 64, 128, 256}) {
        double *a, *b;
        hipMalloc(&a, G * 256 * sizeof(double));
        hipMalloc(&b, G * 256 * sizeof(double));
        hipMemset(a, 0, G * 256 * sizeof(double));
        hipMemset(b, 0, G * 256 * sizeof(double));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        int ph = phases;
        void *args[] = {&a, &b, &ph};
        float ms = 0;
        hipError_t err = hipSuccess;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            err = hipLaunchCooperativeKernel((const void *)k, dim3(G), dim3(256), args, 0, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<double> h(G * 256);
        hipMemcpy(h.data(), (phases % 2) ? b : a, G * 256 * sizeof(double), hipMemcpyDeviceToHost);
        printf("G=%3d: %s, %.1f us for %d phases = %.2f us per phase, value %.0f (expect %d)\n", G,
               hipGetErrorString(err), ms * 1e3, phases, ms * 1e3 / phases, h[5], phases);
        hipFree(a);
        hipFree(b);
    }
    return 0;
}
