// Probe of ds_read_b64_tr_b16: every 16-bit element of LDS holds its own index; lane l passes the address of chunk
// chunk_of[l] (4 elements = 8 bytes); prints which element indices each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* chunk_of, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + chunk_of[threadIdx.x] * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int* d_c; unsigned short* d_o;
    hipMalloc(&d_c, 64 * 4); hipMalloc(&d_o, 256 * 2);
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<int> c(64);
        for (int l = 0; l < 64; ++l) c[l] = variant == 0 ? l : variant == 1 ? l * 5 : (l % 16) * 16 + l / 16;   // identity, stride-5 chunks, [16 rows][4 chunks] image
        hipMemcpy(d_c, c.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_c, d_o);
        std::vector<unsigned short> o(256);
        hipMemcpy(o.data(), d_o, 512, hipMemcpyDeviceToHost);
        printf("variant %d (lane: chunk -> 4 element indices, as chunk.pos)\n", variant);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d chunk %3d :", l, c[l]);
            for (int j = 0; j < 4; ++j) printf(" %3d.%d", o[l * 4 + j] / 4, o[l * 4 + j] % 4);
            printf("\n");
        }
    }
    return 0;
}
