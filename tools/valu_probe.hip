// valu_probe.hip — issue throughput of the integer VALU / cross-lane instructions the
// step kernel is made of, per SIMD, at 1, 2, 4 and 8 waves per SIMD (gfx950).
// Every wave runs REPS x 32 independent instructions of one kind; cycles from s_memtime
// around the loop (median over waves), so the figure is "cycles per wave-instruction per
// SIMD" = wave cycles / (instructions x waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_probe tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REPS 512

#define OP8(s) s s s s s s s s
#define BODY32(s) OP8(s) OP8(s) OP8(s) OP8(s)

template <int KIND>
__global__ __launch_bounds__(256) void probe(uint64_t *cyc, uint32_t *sink)
{
    uint32_t a = threadIdx.x, b = blockIdx.x + 7u, c = 3u, d = 5u;
    uint64_t q = ((uint64_t)a << 32) | b;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if (KIND == 0) { BODY32(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
        if (KIND == 1) { BODY32(asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(c) : "v"(d));) }
        if (KIND == 2) { BODY32(asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c) : "v"(d));) }
        if (KIND == 3) { BODY32(asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q)); asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(q));) }
        if (KIND == 4) { BODY32(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(d));) }
        if (KIND == 5) { BODY32(asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(a) : "v"(b)); asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(c) : "v"(d));) }
        if (KIND == 6) { uint32_t s1, s2; BODY32(asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s1) : "v"(a)); asm volatile("v_readlane_b32 %0, %1, 9" : "=s"(s2) : "v"(c));) a += s1 + s2; }
        if (KIND == 7) { BODY32(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc"); ) }
        if (KIND == 8) { BODY32(asm volatile("v_ffbl_b32 %0, %1" : "=v"(a) : "v"(b)); asm volatile("v_ffbl_b32 %0, %1" : "=v"(c) : "v"(d));) }
        if (KIND == 10) { BODY32(asm volatile("s_nop 0"); asm volatile("s_nop 0");) }
        if (KIND == 11) { BODY32(asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"); asm volatile("s_waitcnt lgkmcnt(0)");) }
        if (KIND == 9) { uint32_t s1 = r, s2 = r + 1; BODY32(asm volatile("s_and_b32 %0, %0, %1" : "+s"(s1) : "s"(s2)); asm volatile("s_or_b32 %0, %0, %1" : "+s"(s2) : "s"(s1));) a += s1 + s2; }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if ((a ^ b ^ c ^ d ^ (uint32_t)q) == 0x12345u) sink[0] = a;
}

template <int KIND>
static void run(const char *name, int per_instr)
{
    uint64_t *cyc;
    uint32_t *sink;
    hipMalloc(&cyc, 8 * 8192 * 4);
    hipMalloc(&sink, 64);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;  // 256-thread blocks: one wave per SIMD each; wps blocks per CU
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, cyc, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double ninstr = (double)REPS * 64 * per_instr / 2.0;   // BODY32 holds 64 statements (32 x 2) or 32
        printf("%-28s waves/SIMD %d: %.2f memtime-ticks per wave-instr per wave, %.2f per SIMD-issue (x%d waves); kernel %.3f ms\n", name, wps,
               med / ninstr, med / ninstr / wps, wps, ms);
    }
    hipFree(cyc); hipFree(sink);
}

// `valu_probe scalar`: only the scalar kinds (for a --pmc SQ_INSTS_SALU pass: do s_nop and s_waitcnt count as SALU? each
// wave issues REPS x 64 of them)
int main(int argc, char **argv)
{
    if (argc > 1) {
        run<9>("s_and/s_or", 2);
        run<10>("s_nop 0", 2);
        run<11>("s_waitcnt", 2);
        return 0;
    }
    run<0>("v_add_u32", 2);
    run<1>("v_and/or_b32", 2);
    run<2>("v_bcnt_u32_b32", 2);
    run<3>("v_lshl/lshr_b64", 2);
    run<4>("v_mov_b32_dpp", 2);
    run<5>("v_bfe/v_lshl_or", 2);
    run<6>("v_readlane_b32", 2);
    run<7>("v_cmp+v_cndmask", 2);
    run<8>("v_ffbl_b32", 2);
    run<9>("s_and/s_or", 2);
    return 0;
}
