#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    int l = threadIdx.x;
    int xi = __float_as_int((float)l);
    out[l] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-1.0f), xi, 0x130, 0xF, 0xF, false));        // wave_shl:1
    out[64 + l] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-1.0f), xi, 0x138, 0xF, 0xF, false));   // wave_shr:1
    out[128 + l] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-1.0f), xi, 0xB1, 0xF, 0xF, false));   // quad_perm 1,0,3,2
    out[192 + l] = __shfl_down((float)l, 1, 64);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"wave_shl1", "wave_shr1", "quad1032", "shfl_down1"};
    for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; ++i) printf(" %g", h[t * 64 + i]); printf("\n"); }
    return 0;
}
