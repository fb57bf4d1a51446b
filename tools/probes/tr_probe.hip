#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int pitch_elems) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    // group g reads rows 4g..4g+3 (pitch given), lane i: row (i>>2), cols (i&3)*4..+3
    const unsigned short* p = lds + (4 * g + (i >> 2)) * pitch_elems + (i & 3) * 4;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int pitch : {16, 40}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pitch);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pitch %d\n", pitch);
        for (int l = 0; l < 64; l += 1) { if (l % 16 < 3 || l % 16 == 15) printf(" lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    }
    return 0;
}
