// calib_traffic.hip - known-byte-count access patterns of the tree loop, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE
// on THIS access pattern (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
// Not part of the product library.   hipcc --offload-arch=gfx950 -O3 -o calib_traffic calib_traffic.hip
//   ./calib_traffic <pattern> <GiB of buffer> <accesses per lane>      (prints the algorithmic bytes it moved)
// Patterns (one wave per workgroup, 4096 workgroups, every lane walks its own pseudo-random sequence over the buffer):
//   rows32   a wave reads ROW consecutive 32-byte records (x4 + x3 loads per lane, like the query's visit) at a random row start
//   rec32    every lane reads one 32-byte record at a random position          (rewire candidates: vrec)
//   rec64    every lane reads one 64-byte record at a random position          (tree records: topo)
//   word4    every lane reads one 4-byte word at a random position             (pos[], queue words)
//   w8       every lane writes 8 bytes at a random position                    (cost / link updates)
//   w32      every lane writes a 32-byte record at a random position           (index rebuild scatter)
//   w64      every lane writes a 64-byte record at a random position           (vertex insertion)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
#define GAS __attribute__((address_space(1)))

__device__ __forceinline__ unsigned long long lcg(unsigned long long &s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return s >> 20;
}

template <int PAT>
__global__ __launch_bounds__(64) void k(char *buf, unsigned long long n_rec32, int iters, unsigned *sink)
{
    const int lane = threadIdx.x;
    unsigned long long sw = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);            // per-wave sequence
    unsigned long long sl = sw ^ (0xD1B54A32D192ED03ull * (unsigned long long)(lane + 1));   // per-lane sequence
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        if (PAT == 0) {   // rows32: 64 consecutive records per trip, row start random (wave-uniform)
            const unsigned long long r0 = lcg(sw) % (n_rec32 - 64);
            const GAS char *p = (const GAS char *)buf + (r0 + (unsigned long long)lane) * 32;
            const v4u a = *(const GAS v4u *)p;
            const v3u b = *(const GAS v3u *)(p + 16);
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z;
        } else if (PAT == 1) {
            const unsigned long long r = lcg(sl) % n_rec32;
            const GAS v4u *p = (const GAS v4u *)((const GAS char *)buf + r * 32);
            const v4u a = p[0], b = p[1];
            acc += a.x + a.w + b.x + b.w;
        } else if (PAT == 2) {
            const unsigned long long r = lcg(sl) % (n_rec32 / 2);
            const GAS v4u *p = (const GAS v4u *)((const GAS char *)buf + r * 64);
            const v4u a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (PAT == 3) {
            const unsigned long long r = lcg(sl) % (n_rec32 * 8);
            acc += *(const GAS unsigned *)((const GAS char *)buf + r * 4);
        } else if (PAT == 4) {
            const unsigned long long r = lcg(sl) % (n_rec32 * 4);
            *(GAS unsigned long long *)((GAS char *)buf + r * 8) = sl;
        } else if (PAT == 5) {
            const unsigned long long r = lcg(sl) % n_rec32;
            GAS v4u *p = (GAS v4u *)((GAS char *)buf + r * 32);
            const v4u v = {(unsigned)sl, (unsigned)it, 1u, 2u};
            p[0] = v; p[1] = v;
        } else {
            const unsigned long long r = lcg(sl) % (n_rec32 / 2);
            GAS v4u *p = (GAS v4u *)((GAS char *)buf + r * 64);
            const v4u v = {(unsigned)sl, (unsigned)it, 1u, 2u};
            p[0] = v; p[1] = v; p[2] = v; p[3] = v;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const char *names[] = {"rows32", "rec32", "rec64", "word4", "w8", "w32", "w64"};
    const int bytes_per_lane[] = {28, 32, 64, 4, 8, 32, 64};
    if (argc < 4) { fprintf(stderr, "usage: %s <pattern> <GiB> <accesses per lane>\n", argv[0]); return 2; }
    int pat = -1;
    for (int i = 0; i < 7; i++) if (!strcmp(argv[1], names[i])) pat = i;
    if (pat < 0) { fprintf(stderr, "unknown pattern\n"); return 2; }
    const double gib = atof(argv[2]);
    const int iters = atoi(argv[3]);
    const size_t bytes = (size_t)(gib * (double)(1ull << 30)) / 4096 * 4096;
    char *buf = nullptr;
    unsigned *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 4096;
    const unsigned long long n32 = bytes / 32;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    switch (pat) {
    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, buf, n32, iters, sink); break;
    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, buf, n32, iters, sink); break;
    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, buf, n32, iters, sink); break;
    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, buf, n32, iters, sink); break;
    case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64), 0, 0, buf, n32, iters, sink); break;
    case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(64), 0, 0, buf, n32, iters, sink); break;
    default: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(64), 0, 0, buf, n32, iters, sink); break;
    }
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)grid * 64.0 * (double)iters * bytes_per_lane[pat];
    printf("{\"pattern\": \"%s\", \"buffer_GiB\": %.2f, \"accesses_per_lane\": %d, \"algorithmic_bytes\": %.0f, \"kernel_ms\": %.3f, \"GBps\": %.1f}\n",
           names[pat], gib, iters, moved, ms, moved / ms / 1e6);
    return 0;
}
