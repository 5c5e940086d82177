// pmc_calib.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for the step
// kernel's access pattern (MI355X_MICROARCH.md "HBM": widths other than 16 B per
// lane are uncalibrated).  Each wave reads R random 128-B rows of a 2 GiB buffer,
// one dword per lane (lane l reads dword l&31), like build_local does, and
// read-modify-writes one dword in every 8th row, like the commit does.
// Known bytes: reads = waves*R*128 (whole line touched), writes = waves*R/8 dwords.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(64) void calib_kernel(uint32_t *buf, uint64_t nrows, int R, uint32_t *sink, int do_write)
{
    const int lane = threadIdx.x;
    uint64_t x = (uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int i = 0; i < R; i += 16) {
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t row = (x >> 20) % nrows;
            w[q] = buf[row * 32 + (lane & 31)];
            if (do_write && q == 0 && lane == 0) buf[row * 32 + 5] = w[q] + 1;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc ^= w[q];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const uint64_t bytes = 2ull << 30;
    const uint64_t nrows = bytes / 128;
    const int waves = argc > 1 ? atoi(argv[1]) : 4096, R = argc > 2 ? atoi(argv[2]) : 4096;
    uint32_t *buf, *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(calib_kernel, dim3(waves), dim3(64), 0, 0, buf, nrows, R, sink, mode);
            hipDeviceSynchronize();
        }
    }
    printf("calib: waves=%d R=%d read_bytes_per_launch=%llu write_dwords_per_launch(mode1)=%llu\n", waves, R,
           (unsigned long long)waves * R * 128ull, (unsigned long long)waves * (R / 16));
    return 0;
}
