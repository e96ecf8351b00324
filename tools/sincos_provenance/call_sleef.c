#include <immintrin.h>
#include <dlfcn.h>
#include <stdio.h>
typedef __m512 (*f16)(__m512);
typedef __m256 (*f8)(__m256);
static void* h;
int init(const char* lib) { h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL); return h != 0; }
int call16(const char* name, const float* in, float* out, long n) {
  f16 f = (f16)dlsym(h, name); if (!f) return -1;
  for (long i = 0; i + 16 <= n; i += 16) _mm512_storeu_ps(out + i, f(_mm512_loadu_ps(in + i)));
  return 0;
}
int call8(const char* name, const float* in, float* out, long n) {
  f8 f = (f8)dlsym(h, name); if (!f) return -1;
  for (long i = 0; i + 8 <= n; i += 8) _mm256_storeu_ps(out + i, f(_mm256_loadu_ps(in + i)));
  return 0;
}
