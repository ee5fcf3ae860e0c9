// Where does workgroup b of a one-round launch run?  Records (XCC, SE, CU) of every workgroup of a grid of 256-thread workgroups
// holding 52 KB of LDS (two per CU, like conv_halo_gb_kernel), each spinning ~30 us so the whole round is resident at once.
// build + run: hipcc --offload-arch=gfx950 -O2 tools/place_probe.hip -o /tmp/place_probe && /tmp/place_probe [nwg]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, long long spin) {
    __shared__ char lds[52 * 1024];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (unsigned)t0; out[blockIdx.x * 4 + 3] = lds[5]; }
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    unsigned* d;
    hipMalloc(&d, n * 16);
    probe<<<n, 256>>>(d, 3000);            // 100 MHz wall clock: 30 us
    hipDeviceSynchronize();
    std::vector<unsigned> h(n * 4);
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    printf("# b xcc se sh cu simd  t0\n");
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 15;
        printf("%4d %2u %u %u %2u %u %u\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, h[b * 4 + 2]);
    }
    return 0;
}
