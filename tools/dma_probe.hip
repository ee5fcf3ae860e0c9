#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void probe(const uint32_t* src, uint32_t* out, unsigned nbytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    int l = threadIdx.x;
    // lane l fetches source chunk (63 - l) (a gather), every 5th lane out of range
    unsigned off = (63 - l) * 16;
    if (l % 5 == 4) off = 0xffffffffu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t h[256], *d, *o; uint32_t ho[1024];
    for (int i = 0; i < 256; ++i) h[i] = i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, o, sizeof(h));
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    printf("before region: %x %x\n", ho[0], ho[255]);
    for (int l = 0; l < 12; ++l) printf("lane %d -> lds dwords: %u %u %u %u\n", l, ho[256 + l * 4], ho[256 + l * 4 + 1], ho[256 + l * 4 + 2], ho[256 + l * 4 + 3]);
    printf("after region: %x\n", ho[512]);
}
