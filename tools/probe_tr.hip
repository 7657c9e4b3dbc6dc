// Hardware probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950 (used to design the wgrad operand gather).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int lane = threadIdx.x;
  // mode 0: lane l -> elements [4l, 4l+4) ; mode 1: row-major [16 rows][pitch 40 elems], lane -> row (l&15), col 4*(l>>4)
  int idx = mode == 0 ? 4 * lane : (lane & 15) * 40 + 4 * (lane >> 4);
  auto r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds + idx));
  short4v s = __builtin_bit_cast(short4v, r);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)s[j];
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, 512);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
