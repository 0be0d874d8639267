// What does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (raw buffer, LDS-DMA) leave in LDS: zeros or the old contents?
// And is the scalar offset part of the range check?   hipcc --offload-arch=gfx950 -O2 buffer_lds_oob_probe.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lptr_t;

__global__ void probe(const uint32_t* src, int bytes, int soff, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t sm[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sm[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    int voff = lane * 16;                       // lanes 0..63 -> bytes 0..1023
    if (lane >= 48) voff = 0x7ffffff0;          // far out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t*)sm, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = sm[i];
}

int main() {
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
    uint32_t *src, *out;
    hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    const int cases[3][2] = {{4096, 0}, {512, 0}, {512, 256}};   // (num_records bytes, scalar offset bytes)
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, c[0], c[1], out);
        std::vector<uint32_t> r(256);
        hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost);
        printf("num_records %d, soffset %d: ", c[0], c[1]);
        for (int l : {0, 15, 16, 31, 32, 47, 48, 63}) printf("lane%d=%08x ", l, r[4 * l]);
        printf("\n");
    }
    return 0;
}
