// dpp_row.hip — which lane does a DPP row shift read?  (hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_row tools/ubench/dpp_row.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, 0xF, 0xF, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = dpp<0x101>(l); out[64 + l] = dpp<0x111>(l); out[128 + l] = dpp<0x121>(l); out[192 + l] = dpp<0x103>(l); out[256 + l] = dpp<0x115>(l);
}
int main() {
    int* d; hipMalloc(&d, 320 * 4); k<<<1, 64>>>(d); int h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* n[5] = {"row_shl:1", "row_shr:1", "row_ror:1", "row_shl:3", "row_shr:5"};
    for (int r = 0; r < 5; ++r) { printf("%-10s lane i reads:", n[r]); for (int i = 0; i < 18; ++i) printf(" %d", h[64 * r + i]); printf("\n"); }
    return 0;
}
