#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;
__global__ void k(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[2048];
    const int lane = threadIdx.x & 63;
    // each lane passes ITS OWN global address (chunk lane), LDS base wave-uniform
    __builtin_amdgcn_global_load_lds((gvoid*)(src + 4 * lane), (lvoid*)buf, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gvoid*)(src + 4 * (lane + 64)), (lvoid*)(buf + 300), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc
    __syncthreads();
    for (int i = lane; i < 600; i += 64) out[i] = buf[i];
}
int main() {
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1000;
    uint32_t *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096); hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipMemset(o, 0, 4096);
    k<<<1, 64>>>(d, o);
    uint32_t r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (r[i] != h[i]) ++bad;
    for (int i = 0; i < 256; ++i) if (r[300 + i] != h[256 + i]) ++bad;
    printf("global_load_lds b128: bad=%d  r[0..3]=%u %u %u %u r[300]=%u r[555]=%u\n", bad, r[0], r[1], r[2], r[3], r[300], r[555]);
    return 0;
}
