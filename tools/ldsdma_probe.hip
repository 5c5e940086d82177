// Probe: where do the bytes of a global->LDS direct load land for sizes 1, 2 and 4?
// (cdna_hip_programming.md: "wave-uniform base + lane x size".)  hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define PROBE(NAME, SIZE)                                                                                   \
    __global__ void NAME(const uint8_t *src, uint8_t *out)                                                  \
    {                                                                                                       \
        extern __shared__ uint8_t lds[];                                                                    \
        const int lane = threadIdx.x;                                                                       \
        for (int i = lane; i < 1024; i += 64) lds[i] = 0xEE;                                                \
        __syncthreads();                                                                                    \
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + 8 * lane),  \
                                         (void __attribute__((address_space(3))) *)(lds + 16), SIZE, 0, 0); \
        __builtin_amdgcn_s_waitcnt(0);                                                                      \
        __syncthreads();                                                                                    \
        for (int i = lane; i < 1024; i += 64) out[i] = lds[i];                                              \
    }
PROBE(k1, 1)
PROBE(k2, 2)
PROBE(k4, 4)
static void show(const char *nm, const uint8_t *r)
{
    printf("%s:", nm);
    for (int i = 0; i < 288; ++i) printf(" %02x", r[i]);
    printf("\n");
}
int main()
{
    uint8_t h[1024], *d, *o, r[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (uint8_t)(i & 0xFF);
    (void)hipMalloc(&d, 1024); (void)hipMalloc(&o, 1024);
    (void)hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    k1<<<1, 64, 1024>>>(d, o); (void)hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost); show("size1", r);
    k2<<<1, 64, 1024>>>(d, o); (void)hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost); show("size2", r);
    k4<<<1, 64, 1024>>>(d, o); (void)hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost); show("size4", r);
    return 0;
}
