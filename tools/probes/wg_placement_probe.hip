// Where does the dispatcher put the workgroups of a grid of ~2 per CU?  548 workgroups of 256 threads holding 33 KB of LDS (the footprint of
// conv_x3_kernel<64,64,...> on the layer-3 shapes) spin for ~20 us and record the XCC / SE / CU they run on; the host prints the histogram of
// workgroups per CU.   hipcc --offload-arch=gfx950 -O2 -o wg_placement_probe wg_placement_probe.hip && ./wg_placement_probe [wgs] [lds bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *out, long spin) {
    extern __shared__ char lds[];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID: xcc_id [3:0]
    long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) lds[threadIdx.x] += 1;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

int main(int argc, char **argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 548, ldsb = argc > 2 ? atoi(argv[2]) : 33 * 1024;
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 2 * wgs);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), ldsb, 0, d, 2000L);   // 100 MHz clock: 20 us
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * wgs);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * wgs, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    for (int i = 0; i < wgs; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
        per_cu[(xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)]++;
    }
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("%d workgroups with %d B of LDS: %zu distinct (xcc, se, sh, cu) slots used;", wgs, ldsb, per_cu.size());
    for (auto &kv : hist) printf("  %d CUs hold %d", kv.second, kv.first);
    printf("\n");
    return 0;
}
