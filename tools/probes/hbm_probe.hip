// HBM ceiling of this device measured with hand-written 16-byte-lane kernels (tools/hbm_probe.py timed ATen's
// copy_ / fill_, which are not the ceiling), plus the access shape of the residual-tail kernels: persistent workgroups that read PIECE
// bytes out of every row of a tile per step (rows PITCH bytes apart) and walk along the rows step by step.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_probe.hip -o tools/probes/hbm_probe.bin     Run: ./hbm_probe.bin [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n, int nt) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {        // four 16-byte pieces per lane in flight
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i + k * 256 < n ? (nt ? __builtin_nontemporal_load(src + i + k * 256) : src[i + k * 256]) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 256 < n) { if (nt) __builtin_nontemporal_store(v[k], dst + i + k * 256); else dst[i + k * 256] = v[k]; }
    }
}
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *__restrict__ src, unsigned *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i + k * 256 < n ? src[i + k * 256] : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
    }
    if (acc == 0x12345678u) out[0] = acc;                                                       // (never: keeps the loads alive)
}
__global__ __launch_bounds__(256) void write_kernel(u32x4 *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 256 < n) dst[i + k * 256] = v;
}
// the tail kernels' shape: a workgroup owns tiles of 128 rows (PITCH bytes each); step s reads bytes [s * PIECE, (s + 1) * PIECE) of each of
// the 128 rows of TWO tensors and writes the same range of a third; DEPTH steps' loads in flight (registers)
template <int PIECE, int DEPTH>
__global__ __launch_bounds__(512) void piece_kernel(const unsigned char *__restrict__ a, const unsigned char *__restrict__ b, unsigned char *__restrict__ c,
                                                    int ntiles, int pitch, int wr) {
    constexpr int LPR = PIECE / 16, RPP = 512 / LPR, NJ = 128 / RPP;                           // lanes per row, rows per pass, passes
    static_assert(NJ >= 1 && NJ * DEPTH <= 16, "register budget");
    const int l = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
    const int steps = pitch / PIECE;
    for (int L = blockIdx.x; L < ntiles; L += gridDim.x) {
        const size_t base = (size_t)L * 128 * pitch;
        u32x4 va[DEPTH][NJ], vb[DEPTH][NJ];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const size_t off = base + (size_t)(r0 + RPP * j) * pitch + (size_t)d * PIECE + l * 16;
                va[d][j] = *reinterpret_cast<const u32x4 *>(a + off); vb[d][j] = *reinterpret_cast<const u32x4 *>(b + off);
            }
        for (int s0 = 0; s0 < steps; s0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int s = s0 + d;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const size_t off = base + (size_t)(r0 + RPP * j) * pitch + (size_t)s * PIECE + l * 16;
                    const u32x4 o = va[d][j] ^ vb[d][j];
                    if (wr) *reinterpret_cast<u32x4 *>(c + off) = o;
                    else if (o[0] == 0x12345678u) *reinterpret_cast<u32x4 *>(c + off) = o;
                }
                const int sn = s + DEPTH < steps ? s + DEPTH : s;                               // (past the end: a harmless re-read)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const size_t off = base + (size_t)(r0 + RPP * j) * pitch + (size_t)sn * PIECE + l * 16;
                    va[d][j] = *reinterpret_cast<const u32x4 *>(a + off); vb[d][j] = *reinterpret_cast<const u32x4 *>(b + off);
                }
            }
        }
    }
}

static double time_it(void (*launch)(void *), void *ctx, int reps = 7) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(ctx); hipDeviceSynchronize();
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0); launch(ctx); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e-3);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

struct Ctx { unsigned char *a, *b, *c; size_t bytes; int grid, nt, pitch, ntiles, wr; };

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1024;
    const size_t bytes = mib << 20;
    Ctx c{};
    hipMalloc(&c.a, bytes); hipMalloc(&c.b, bytes); hipMalloc(&c.c, bytes);
    hipMemset(c.a, 1, bytes); hipMemset(c.b, 2, bytes); hipMemset(c.c, 3, bytes);
    c.bytes = bytes;
    printf("buffers of %zu MiB (Infinity Cache: 256 MiB); median of 7 launches; TB = 1e12 bytes\n", mib);
    for (int grid : {2048, 8192, 65536}) {
        c.grid = grid;
        for (int nt = 0; nt < 2; ++nt) {
            c.nt = nt;
            const double t = time_it([](void *p) { Ctx *x = (Ctx *)p; hipLaunchKernelGGL(copy_kernel, dim3(x->grid), dim3(256), 0, 0, (const u32x4 *)x->a, (u32x4 *)x->c, x->bytes / 16, x->nt); }, &c);
            printf("copy  grid %6d %s: %6.2f TB/s (read + write)\n", grid, nt ? "nontemporal" : "plain      ", 2.0 * bytes / t / 1e12);
        }
        double t = time_it([](void *p) { Ctx *x = (Ctx *)p; hipLaunchKernelGGL(read_kernel, dim3(x->grid), dim3(256), 0, 0, (const u32x4 *)x->a, (unsigned *)x->c, x->bytes / 16); }, &c);
        printf("read  grid %6d            : %6.2f TB/s\n", grid, bytes / t / 1e12);
        t = time_it([](void *p) { Ctx *x = (Ctx *)p; hipLaunchKernelGGL(write_kernel, dim3(x->grid), dim3(256), 0, 0, (u32x4 *)x->c, x->bytes / 16); }, &c);
        printf("write grid %6d            : %6.2f TB/s\n", grid, bytes / t / 1e12);
    }
    // tail-kernel shape: 134 MB tensors as the layer3 block output at config C2 (65,536 rows of 2 KiB), 256 / 512 persistent workgroups
    for (int pitch : {2048, 1024, 4096}) {
        const size_t tb = (size_t)65536 * 2048;                                                  // bytes per tensor
        c.pitch = pitch; c.ntiles = (int)(tb / pitch / 128);
        for (int wr = 0; wr < 2; ++wr) {
            c.wr = wr;
            for (int grid : {256}) {
                c.grid = grid;
                const double traffic = (wr ? 3.0 : 2.0) * tb;
#define RUN(P, D)                                                                                                                             \
    {                                                                                                                                         \
        const double t = time_it([](void *p) { Ctx *x = (Ctx *)p; hipLaunchKernelGGL((piece_kernel<P, D>), dim3(x->grid), dim3(512), 0, 0, x->a, x->b, x->c, x->ntiles, x->pitch, x->wr); }, &c); \
        printf("pieces pitch %4d piece %4d depth %d grid %3d %s: %6.2f TB/s  (%.0f us)\n", pitch, P, D, grid, wr ? "2 reads + 1 write" : "2 reads          ", traffic / t / 1e12, t * 1e6); \
    }
                RUN(64, 4) RUN(64, 8) RUN(128, 2) RUN(128, 4) RUN(128, 8) RUN(256, 2) RUN(256, 4) RUN(512, 2) RUN(1024, 1)
            }
        }
    }
    return 0;
}
