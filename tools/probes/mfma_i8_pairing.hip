// Which element (lane half, byte) of the A fragment meets which element of the B fragment in v_mfma_i32_32x32x32_i8?
// A = 1 at element (ha, ja) of every row, B = 1 at element (hb, jb) of every column: D = 1 iff both are the same k.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void probe(int* out)
{
    const int lane = threadIdx.x, hh = lane >> 5;
    for (int ea = 0; ea < 32; ea++)
        for (int eb = 0; eb < 32; eb++) {
            v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
            if (hh == (ea >> 4)) a[(ea & 15) >> 2] = 1 << (8 * (ea & 3));
            if (hh == (eb >> 4)) b[(eb & 15) >> 2] = 1 << (8 * (eb & 3));
            v16i c = {};
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
            if (lane == 0) out[ea * 32 + eb] = c[0];
        }
}
int main()
{
    int* d;
    hipMalloc(&d, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    int h[1024];
    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    for (int ea = 0; ea < 32; ea++) {
        printf("A elem (h%d,j%2d) pairs with B:", ea >> 4, ea & 15);
        for (int eb = 0; eb < 32; eb++)
            if (h[ea * 32 + eb]) printf(" (h%d,j%d)=%d", eb >> 4, eb & 15, h[ea * 32 + eb]);
        printf("\n");
    }
    return 0;
}
