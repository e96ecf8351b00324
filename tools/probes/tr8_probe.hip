#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((vector_size(8)));
__global__ void k(unsigned* o, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char L[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) L[i] = (unsigned char)(i & 0xff);
  __syncthreads();
  // each lane supplies the address of 8 contiguous bytes: lane l -> byte offset 8*l (+ 256*page so values are unique mod 256 per 32 lanes)
  const int l = threadIdx.x;
  int off = mode == 0 ? 8 * l : 16 * l;   // mode 1: stride 16 (only the low 8 bytes of every 16)
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(L + off));
  o[2 * l] = r[0]; o[2 * l + 1] = r[1];
}
int main() {
  unsigned* o; (void)hipMalloc(&o, 4096);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(o, mode);
    unsigned ho[128]; (void)hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    printf("mode %d (lane l supplies bytes at %s; value = byte offset & 255)\n", mode, mode ? "16*l" : "8*l");
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int b = 0; b < 8; ++b) printf(" %3u", (ho[2*l + b/4] >> (8*(b%4))) & 0xff); printf("\n"); }
  }
  return 0;
}
