// Probe: what does an LDS-DMA buffer load (buffer_load_dwordx4 ... offen lds) write for a lane whose offset is
// outside the descriptor's num_records?  (zeros, or nothing = stale LDS bytes).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
__global__ void k(const unsigned char *x, int nbytes, const unsigned *offs, unsigned *out) {
    __shared__ uint4 lds[256];
    lds[threadIdx.x] = make_uint4(0xABABABABu, 0xABABABABu, 0xABABABABu, 0xABABABABu);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + wave * 64), 16, offs[threadIdx.x], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4 v = lds[threadIdx.x];
    out[4 * threadIdx.x + 0] = v.x; out[4 * threadIdx.x + 1] = v.y; out[4 * threadIdx.x + 2] = v.z; out[4 * threadIdx.x + 3] = v.w;
}
int main() {
    const int n = 4096;
    std::vector<unsigned char> h(n);
    for (int i = 0; i < n; ++i) h[i] = (unsigned char)(i * 7 + 1);
    std::vector<unsigned> offs(256);
    for (int i = 0; i < 256; ++i) offs[i] = (i % 3 == 0) ? 0xFFFFFFF0u : (i % 3 == 1 ? 0x80000000u : (unsigned)(i * 16) % (n - 16));
    unsigned char *dx; unsigned *doffs, *dout;
    hipMalloc(&dx, n); hipMalloc(&doffs, 1024); hipMalloc(&dout, 4096);
    hipMemcpy(dx, h.data(), n, hipMemcpyHostToDevice); hipMemcpy(doffs, offs.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dx, n, doffs, dout);
    std::vector<unsigned> o(1024);
    hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    int zeros = 0, stale = 0, other = 0, good = 0, bad = 0;
    for (int i = 0; i < 256; ++i) {
        if (i % 3 != 2) {
            bool z = true, s = true;
            for (int j = 0; j < 4; ++j) { z &= o[4 * i + j] == 0; s &= o[4 * i + j] == 0xABABABABu; }
            zeros += z; stale += s; other += !z && !s;
        } else {
            unsigned e; memcpy(&e, h.data() + offs[i], 4);
            (o[4 * i] == e ? good : bad)++;
        }
    }
    printf("lds-dma OOB lanes: zeros=%d stale=%d other=%d ; in-range lanes good=%d bad=%d\n", zeros, stale, other, good, bad);
    return 0;
}
